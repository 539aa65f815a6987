"""Search-space domains and the variant resolver.

A search space is a plain (possibly nested) dict whose leaves are either constants or
``Domain`` objects.  ``resolve(space, rng)`` turns it into one concrete trial config.

Covers the domain kinds the reference scripts use --
``tune.choice`` / ``tune.uniform`` / ``tune.loguniform`` / ``tune.sample_from``
(/root/reference/ray-tune-hpo-regression.py:379-400,
/root/reference/ray-tune-hpo-regression-sample.py:140-147) -- plus ``randint``,
``quniform``, ``qloguniform``, ``randn`` and ``grid_search``.

Differences from the reference's behaviour that are deliberate:

* ``sample_from`` lambdas are resolved after the keys they read (dependency-ordered retry),
  and a lambda may itself return a ``Domain`` which is then sampled.  That is what makes the
  reference's ``dim_feedforward = d_model * choice([2, 3, 4])`` intent
  (ray-tune-hpo-regression.py:383) expressible: ``sample_from(lambda spec:
  spec.config.d_model * choice([2, 3, 4]))`` works because ``Domain.__rmul__`` builds a
  derived domain instead of raising ``TypeError``.
* every draw comes from an explicit ``numpy.random.Generator`` so an experiment is
  reproducible from one integer seed on every rank of a multi-GPU job.
"""
from __future__ import annotations

import copy
import itertools
import math
from typing import Any, Callable, Dict, Iterator, List, Sequence, Tuple

import numpy as np


class Domain:
    """Base class of all sampleable hyper-parameter domains."""

    def sample(self, rng: np.random.Generator) -> Any:  # pragma: no cover - abstract
        raise NotImplementedError

    # --- arithmetic on domains builds derived domains (used inside sample_from lambdas) ---
    def _derived(self, fn: Callable[[Any], Any]) -> "Derived":
        return Derived(self, fn)

    def __mul__(self, other):
        return self._derived(lambda v: v * other)

    __rmul__ = __mul__

    def __add__(self, other):
        return self._derived(lambda v: v + other)

    __radd__ = __add__

    def __sub__(self, other):
        return self._derived(lambda v: v - other)

    def __rsub__(self, other):
        return self._derived(lambda v: other - v)

    def __truediv__(self, other):
        return self._derived(lambda v: v / other)

    def __floordiv__(self, other):
        return self._derived(lambda v: v // other)

    # --- introspection used by model-based searchers ---
    @property
    def is_continuous(self) -> bool:
        return False

    def to_unit(self, value: Any) -> float:
        """Map a value of this domain to [0, 1] (searcher encoding)."""
        raise NotImplementedError

    def from_unit(self, u: float) -> Any:
        """Inverse of ``to_unit`` (with rounding / snapping where the domain is discrete)."""
        raise NotImplementedError


class Derived(Domain):
    """A domain obtained by applying a function to another domain's sample."""

    def __init__(self, base: Domain, fn: Callable[[Any], Any]):
        self.base, self.fn = base, fn

    def sample(self, rng):
        return self.fn(self.base.sample(rng))

    def __repr__(self):
        return f"Derived({self.base!r})"


class Categorical(Domain):
    def __init__(self, categories: Sequence[Any]):
        cats = list(categories)
        if not cats:
            raise ValueError("choice() needs at least one category")
        self.categories = cats

    def sample(self, rng):
        return self.categories[int(rng.integers(len(self.categories)))]

    def to_unit(self, value):
        idx = self.categories.index(value)
        n = len(self.categories)
        return (idx + 0.5) / n

    def from_unit(self, u):
        n = len(self.categories)
        return self.categories[min(n - 1, max(0, int(math.floor(u * n))))]

    def __len__(self):
        return len(self.categories)

    def __repr__(self):
        return f"choice({self.categories!r})"


class Float(Domain):
    """Real interval with a uniform, log-uniform or normal sampler, optionally quantised."""

    def __init__(self, lower: float, upper: float, log: bool = False, q: float | None = None,
                 normal: Tuple[float, float] | None = None):
        if normal is None:
            if not (upper > lower):
                raise ValueError(f"need lower < upper, got [{lower}, {upper}]")
            if log and lower <= 0:
                raise ValueError("loguniform needs a positive lower bound")
        self.lower, self.upper, self.log, self.q, self.normal = lower, upper, log, q, normal

    @property
    def is_continuous(self):
        return self.q is None

    def _snap(self, v: float) -> float:
        if self.q:
            v = round(v / self.q) * self.q
            v = min(max(v, self.lower), self.upper)
        return float(v)

    def sample(self, rng):
        if self.normal is not None:
            mu, sd = self.normal
            return float(rng.normal(mu, sd))
        if self.log:
            v = math.exp(rng.uniform(math.log(self.lower), math.log(self.upper)))
        else:
            v = rng.uniform(self.lower, self.upper)
        return self._snap(v)

    def to_unit(self, value):
        if self.log:
            return (math.log(value) - math.log(self.lower)) / (math.log(self.upper) - math.log(self.lower))
        return (value - self.lower) / (self.upper - self.lower)

    def from_unit(self, u):
        u = min(1.0, max(0.0, float(u)))
        if self.log:
            v = math.exp(math.log(self.lower) + u * (math.log(self.upper) - math.log(self.lower)))
        else:
            v = self.lower + u * (self.upper - self.lower)
        return self._snap(v)

    def __repr__(self):
        kind = "loguniform" if self.log else "uniform"
        return f"{kind}({self.lower}, {self.upper})"


class Integer(Domain):
    """Integers in ``[lower, upper)`` (upper exclusive, like ``numpy.random.randint``)."""

    def __init__(self, lower: int, upper: int, log: bool = False):
        if not (upper > lower):
            raise ValueError("need lower < upper")
        self.lower, self.upper, self.log = int(lower), int(upper), log

    def sample(self, rng):
        if self.log:
            v = math.exp(rng.uniform(math.log(self.lower), math.log(self.upper)))
            return int(min(self.upper - 1, max(self.lower, math.floor(v))))
        return int(rng.integers(self.lower, self.upper))

    def to_unit(self, value):
        return (value - self.lower + 0.5) / (self.upper - self.lower)

    def from_unit(self, u):
        n = self.upper - self.lower
        return self.lower + min(n - 1, max(0, int(math.floor(u * n))))

    def __repr__(self):
        return f"randint({self.lower}, {self.upper})"


class Function(Domain):
    """``sample_from(fn)``: ``fn(spec)`` sees the partially resolved config as ``spec.config``."""

    def __init__(self, fn: Callable[[Any], Any]):
        self.fn = fn

    def sample(self, rng, spec=None):
        return self.fn(spec)

    def __repr__(self):
        return "sample_from(<fn>)"


class GridSearch:
    """Marker for an exhaustive axis: every value appears once per ``num_samples`` repeat."""

    def __init__(self, values: Sequence[Any]):
        self.values = list(values)

    def __repr__(self):
        return f"grid_search({self.values!r})"


# ----------------------------------------------------------------------------- constructors
def choice(categories: Sequence[Any]) -> Categorical:
    return Categorical(categories)


def uniform(lower: float, upper: float) -> Float:
    return Float(lower, upper)


def loguniform(lower: float, upper: float) -> Float:
    return Float(lower, upper, log=True)


def quniform(lower: float, upper: float, q: float) -> Float:
    return Float(lower, upper, q=q)


def qloguniform(lower: float, upper: float, q: float) -> Float:
    return Float(lower, upper, log=True, q=q)


def randn(mean: float = 0.0, sd: float = 1.0) -> Float:
    return Float(-math.inf, math.inf, normal=(mean, sd))


def randint(lower: int, upper: int) -> Integer:
    return Integer(lower, upper)


def lograndint(lower: int, upper: int) -> Integer:
    return Integer(lower, upper, log=True)


def sample_from(fn: Callable[[Any], Any]) -> Function:
    return Function(fn)


def grid_search(values: Sequence[Any]) -> GridSearch:
    return GridSearch(values)


# ----------------------------------------------------------------------------- resolver
class _AttrDict(dict):
    """dict with attribute access; reading a still-unresolved key raises ``_Unresolved``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, (Domain, GridSearch)):
            raise _Unresolved(k)
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default


class _Unresolved(Exception):
    pass


class _Spec:
    def __init__(self, config):
        self.config = _AttrDict(config)

    def __getitem__(self, k):  # ``spec["config"]["x"]`` also works
        if k == "config":
            return self.config
        raise KeyError(k)


def _walk(d: Dict, prefix: Tuple = ()) -> Iterator[Tuple[Tuple, Any]]:
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _walk(v, prefix + (k,))
        else:
            yield prefix + (k,), v


def _set(d: Dict, path: Tuple, value: Any) -> None:
    for k in path[:-1]:
        d = d[k]
    d[path[-1]] = value


def has_unresolved(config: Dict) -> bool:
    return any(isinstance(v, (Domain, GridSearch)) for _, v in _walk(config))


def domains(space: Dict) -> Dict[Tuple, Domain]:
    """Flat ``{path: Domain}`` view of the sampleable leaves of a search space."""
    return {p: v for p, v in _walk(space) if isinstance(v, Domain)}


def grid_axes(space: Dict) -> List[Tuple[Tuple, List[Any]]]:
    return [(p, v.values) for p, v in _walk(space) if isinstance(v, GridSearch)]


def _copy_containers(node: Any) -> Any:
    """Copy of the dict/list skeleton of a search space; leaves (domains, scalars, user objects) are shared.
    A full deepcopy per trial was a third of the scheduling cost of a 2048-trial search."""
    if isinstance(node, dict):
        return {k: _copy_containers(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_copy_containers(v) for v in node]
    if isinstance(node, tuple):
        return tuple(_copy_containers(v) for v in node)
    return node


class _Plan:
    """What ``resolve`` finds out about a search space by walking it, kept by ``generate_variants`` for all its draws
    (every rank of a multi-GPU search draws every trial, so this walk was paid trials x ranks times)."""
    __slots__ = ("plain", "functions", "flat")

    def __init__(self, space: Dict):
        self.plain: List[Tuple[Tuple, Domain]] = []
        self.functions: List[Tuple[Tuple, Function]] = []
        for path, leaf in _walk(space):
            if isinstance(leaf, GridSearch):
                raise ValueError(f"grid_search at {path} needs a grid assignment; use generate_variants()")
            if isinstance(leaf, Function):
                self.functions.append((path, leaf))
            elif isinstance(leaf, Domain):
                self.plain.append((path, leaf))
        self.flat = not any(isinstance(v, (dict, list, tuple)) for v in space.values())


def resolve(space: Dict, rng: np.random.Generator, grid_assignment: Dict[Tuple, Any] | None = None,
            plan: "_Plan | None" = None) -> Dict:
    """Sample one concrete config from ``space``.

    Plain domains are drawn in dict order (so a fixed seed gives a fixed config);
    ``sample_from`` leaves are retried until every key they read has been resolved, which
    gives dependency-ordered evaluation without declaring the dependencies.
    """
    pending: List[Tuple[Tuple, Function]]
    if plan is not None and not grid_assignment:
        if plan.flat:
            cfg = dict(space)
            for (key,), leaf in plan.plain:
                cfg[key] = leaf.sample(rng)
        else:
            cfg = _copy_containers(space)
            for path, leaf in plan.plain:
                _set(cfg, path, leaf.sample(rng))
        pending = list(plan.functions)
    else:
        cfg = _copy_containers(space)
        if grid_assignment:
            for p, v in grid_assignment.items():
                _set(cfg, p, v)
        pending = []
        for path, leaf in list(_walk(cfg)):
            if isinstance(leaf, GridSearch):
                raise ValueError(f"grid_search at {path} needs a grid assignment; use generate_variants()")
            if isinstance(leaf, Function):
                pending.append((path, leaf))
            elif isinstance(leaf, Domain):
                _set(cfg, path, leaf.sample(rng))
    while pending:
        still: List[Tuple[Tuple, Function]] = []
        for path, fn in pending:
            try:
                val = fn.sample(rng, _Spec(cfg))
            except _Unresolved:
                still.append((path, fn))
                continue
            # a lambda may hand back a domain (e.g. ``d_model * choice([2, 3, 4])``)
            depth = 0
            while isinstance(val, Domain) and not isinstance(val, Function):
                val = val.sample(rng)
                depth += 1
                if depth > 16:
                    raise RecursionError(f"sample_from at {path} keeps returning domains")
            _set(cfg, path, val)
        if len(still) == len(pending):
            keys = [".".join(map(str, p)) for p, _ in still]
            raise ValueError(f"circular sample_from dependencies among {keys}")
        pending = still
    return cfg


def generate_variants(space: Dict, num_samples: int, seed: int | None = 0) -> Iterator[Dict]:
    """``num_samples`` random draws, multiplied by the cross product of any ``grid_search`` axes."""
    rng = np.random.default_rng(seed)
    axes = grid_axes(space)
    plan = None if axes else _Plan(space)
    for _ in range(num_samples):
        if axes:
            for combo in itertools.product(*[vals for _, vals in axes]):
                yield resolve(space, rng, {p: v for (p, _), v in zip(axes, combo)})
        else:
            yield resolve(space, rng, plan=plan)


def count_variants(space: Dict, num_samples: int) -> int:
    n = num_samples
    for _, vals in grid_axes(space):
        n *= len(vals)
    return n


def flatten(config: Dict, sep: str = "/") -> Dict[str, Any]:
    return {sep.join(map(str, p)): v for p, v in _walk(config)}
