"""Gaussian-process / UCB Bayesian optimisation searcher.

Stands in for ``BayesOptSearch(metric, mode, random_search_steps=10)``
(/root/reference/ray-tune-hpo-regression.py:474): a Matern-5/2 GP with an upper-confidence-bound
acquisition (kappa = 2.576) after ``random_search_steps`` random points.  The reference's
searcher accepts continuous domains only, so on that script's mostly categorical space (and its
``sample_from`` leaf) it fails at space conversion; this implementation encodes every leaf:

* continuous / log / integer domains  -> their position in [0, 1] (log domains in log space),
* numeric categoricals                -> ordinal position (neighbouring sizes are similar),
* other categoricals (strings, bools) -> one-hot block (all categories equidistant),
* ``sample_from`` leaves              -> not encoded; they are functions of the other leaves.

Candidates are drawn *from the search space itself* (so dependent parameters stay consistent) and
ranked by the acquisition; the GP is ~60 lines of numpy/scipy, no external BO package.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.linalg import cho_factor, cho_solve

from .searcher import Searcher
from ..tune import space as sp
from ..tune.space import Categorical, Domain, Float, Function, Integer


# ------------------------------------------------------------------------------- tiny GP
def matern52(a: np.ndarray, b: np.ndarray, length: float) -> np.ndarray:
    d2 = np.maximum(((a[:, None, :] - b[None, :, :]) ** 2).sum(-1), 0.0)
    r = np.sqrt(d2) / length
    s5 = math.sqrt(5.0)
    return (1.0 + s5 * r + (5.0 / 3.0) * r * r) * np.exp(-s5 * r)


class GaussianProcess:
    """Zero-mean GP on standardised targets; isotropic length-scale chosen by marginal likelihood."""

    def __init__(self, noise: float = 1e-6, length_grid: Sequence[float] = (0.05, 0.1, 0.2, 0.35, 0.6, 1.0, 2.0)):
        self.noise, self.length_grid = noise, tuple(length_grid)
        self.x: Optional[np.ndarray] = None

    def fit(self, x: np.ndarray, y: np.ndarray) -> "GaussianProcess":
        self.x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        self.y_mean = float(y.mean())
        self.y_std = float(y.std()) or 1.0
        z = (y - self.y_mean) / self.y_std
        best = None
        for ell in self.length_grid:
            k = matern52(self.x, self.x, ell) + (self.noise + 1e-8) * np.eye(len(z))
            try:
                c = cho_factor(k, lower=True)
            except np.linalg.LinAlgError:
                continue
            alpha = cho_solve(c, z)
            lml = -0.5 * float(z @ alpha) - float(np.log(np.diag(c[0])).sum()) - 0.5 * len(z) * math.log(2 * math.pi)
            if best is None or lml > best[0]:
                best = (lml, ell, c, alpha)
        if best is None:
            raise np.linalg.LinAlgError("GP kernel matrix is not positive definite for any length-scale")
        self.lml, self.length, self._chol, self._alpha = best
        return self

    def predict(self, xs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        xs = np.asarray(xs, dtype=np.float64)
        ks = matern52(xs, self.x, self.length)
        mu = ks @ self._alpha
        v = cho_solve(self._chol, ks.T)
        var = np.maximum(1.0 - (ks * v.T).sum(-1), 1e-12)
        return mu * self.y_std + self.y_mean, np.sqrt(var) * self.y_std


# ------------------------------------------------------------------------------- encoding
class SpaceEncoder:
    def __init__(self, space: Dict):
        self.leaves: List[Tuple[Tuple, Domain, str, int]] = []   # (path, domain, kind, width)
        for path, dom in sp.domains(space).items():
            if isinstance(dom, Function) or not isinstance(dom, (Categorical, Float, Integer)):
                continue
            if isinstance(dom, Float) and dom.normal is not None:
                continue
            if isinstance(dom, Categorical):
                numeric = all(isinstance(c, (int, float)) and not isinstance(c, bool) for c in dom.categories)
                if numeric and len(dom.categories) > 1:
                    self.leaves.append((path, dom, "ordinal", 1))
                elif len(dom.categories) > 1:
                    self.leaves.append((path, dom, "onehot", len(dom.categories)))
            else:
                self.leaves.append((path, dom, "unit", 1))
        self.dim = sum(w for *_, w in self.leaves)

    @staticmethod
    def _get(cfg: Dict, path: Tuple) -> Any:
        for k in path:
            cfg = cfg[k]
        return cfg

    def encode(self, cfg: Dict) -> np.ndarray:
        out = np.zeros(self.dim)
        i = 0
        for path, dom, kind, w in self.leaves:
            v = self._get(cfg, path)
            if kind == "unit":
                out[i] = dom.to_unit(v)
            elif kind == "ordinal":
                order = sorted(dom.categories)
                out[i] = order.index(v) / (len(order) - 1)
            else:
                # scaled so two different categories sit at distance 1, like two ends of an axis
                out[i + dom.categories.index(v)] = 1.0 / math.sqrt(2.0)
            i += w
        return out


class BayesOptSearch(Searcher):
    def __init__(self, space: Optional[Dict] = None, metric: Optional[str] = None, mode: Optional[str] = None,
                 random_search_steps: int = 10, utility_kwargs: Optional[Dict] = None, seed: int = 0,
                 num_candidates: int = 2048, max_suggestions: Optional[int] = None):
        super().__init__(metric, mode)
        self.random_search_steps = random_search_steps
        kw = {"kind": "ucb", "kappa": 2.576, "xi": 0.0}
        kw.update(utility_kwargs or {})
        self.kappa = float(kw["kappa"])
        self.num_candidates = num_candidates
        self.max_suggestions = max_suggestions
        self._rng = np.random.default_rng(seed)
        self._live: Dict[str, Dict] = {}
        self._x: List[np.ndarray] = []
        self._y: List[float] = []
        self._n_suggested = 0
        self._encoder: Optional[SpaceEncoder] = None
        if space is not None:
            self.set_search_properties(metric, mode, space)

    def set_search_properties(self, metric, mode, config) -> bool:
        super().set_search_properties(metric, mode, config)
        if sp.grid_axes(config):
            raise ValueError("grid_search axes cannot be combined with BayesOptSearch")
        self._encoder = SpaceEncoder(config)
        return True

    # ------------------------------------------------------------------ suggest / observe
    def suggest(self, trial_id: str):
        if self._space is None:
            raise RuntimeError("BayesOptSearch has no search space yet")
        if self.max_suggestions is not None and self._n_suggested >= self.max_suggestions:
            return Searcher.FINISHED
        self._n_suggested += 1
        enc = self._encoder
        if len(self._y) < max(1, self.random_search_steps) or enc.dim == 0:
            cfg = sp.resolve(self._space, self._rng)
        else:
            cands = [sp.resolve(self._space, self._rng) for _ in range(self.num_candidates)]
            xs = np.stack([enc.encode(c) for c in cands])
            gp = GaussianProcess().fit(np.stack(self._x), np.asarray(self._y))
            mu, sd = gp.predict(xs)
            # points already running are penalised so parallel suggestions spread out
            if self._live:
                pend = np.stack([enc.encode(c) for c in self._live.values()])
                dist = np.sqrt(((xs[:, None, :] - pend[None, :, :]) ** 2).sum(-1)).min(1)
                sd = sd * np.minimum(1.0, dist / max(gp.length, 1e-9))
            cfg = cands[int(np.argmax(mu + self.kappa * sd))]
        self._live[trial_id] = cfg
        return cfg

    def on_trial_complete(self, trial_id: str, result: Optional[Dict] = None, error: bool = False) -> None:
        cfg = self._live.pop(trial_id, None)
        if cfg is None or error or not result or self.metric not in result:
            return
        v = result[self.metric]
        try:
            v = float(v)
        except (TypeError, ValueError):
            return
        if not math.isfinite(v):      # a diverged trial (nan, +-inf) would turn the GP's target scaling, and with it every
            return                    # later acquisition value, into nan
        self.register(cfg, v)

    def register(self, cfg: Dict, value: float) -> None:
        """Add an observation (the searcher maximises internally; mode=min negates).  Non-finite values are dropped."""
        if not math.isfinite(value):
            return
        self._x.append(self._encoder.encode(cfg))
        self._y.append(-value if self.mode == "min" else value)

    def state_dict(self) -> Dict:
        return {"x": [x.tolist() for x in self._x], "y": list(self._y), "n_suggested": self._n_suggested,
                "rng": self._rng.bit_generator.state}

    def load_state_dict(self, state: Dict) -> None:
        self._x = [np.asarray(x) for x in state.get("x", [])]
        self._y = list(state.get("y", []))
        self._n_suggested = state.get("n_suggested", 0)
        if "rng" in state:
            self._rng.bit_generator.state = state["rng"]
