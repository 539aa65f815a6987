"""Population Based Training.

Not present in the reference (its scripts only build ASHA,
/root/reference/ray-tune-hpo-regression.py:473); required here for the bf16 FT-Transformer
target configuration, where a population of packed trials periodically *exploits* (a
bottom-quantile member copies the weights + optimiser state of a top-quantile member) and
*explores* (perturbs the copied hyper-parameters).

The scheduler only decides *who copies whom and with which new config*; moving the weights is the
executor's job (device-to-device slab copy inside a pack, RCCL send/recv across GPUs, or a
checkpoint hand-over for function trainables).
"""
from __future__ import annotations

import copy
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .base import TrialScheduler, CONTINUE, RESTART, is_nan
from ..tune.space import Domain, Categorical, Float, Integer
from ..tune.trial import Trial


def explore(config: Dict, mutations: Dict[str, Any], resample_probability: float,
            rng: np.random.Generator, perturbation_factors: Tuple[float, float] = (1.2, 0.8),
            custom_explore_fn: Optional[Callable[[Dict], Dict]] = None) -> Dict:
    """Return a perturbed copy of ``config`` for the keys listed in ``mutations``.

    ``mutations[key]`` may be a ``Domain`` (resampled with ``resample_probability``, otherwise a
    numeric value is scaled by 1.2 / 0.8 and clipped to the domain), a list of allowed values
    (resample, or step to a neighbouring entry) or a zero-argument callable (always resampled).
    """
    new = copy.deepcopy(config)
    for key, spec in mutations.items():
        cur = new.get(key)
        if isinstance(spec, dict):
            new[key] = explore(cur or {}, spec, resample_probability, rng, perturbation_factors)
            continue
        if isinstance(spec, Categorical):
            spec = list(spec.categories)
        if isinstance(spec, (list, tuple)):
            values = list(spec)
            if rng.random() < resample_probability or cur not in values:
                new[key] = values[int(rng.integers(len(values)))]
            else:
                i = values.index(cur)
                step = 1 if rng.random() > 0.5 else -1
                new[key] = values[min(len(values) - 1, max(0, i + step))]
        elif isinstance(spec, Domain):
            if rng.random() < resample_probability or not isinstance(cur, (int, float)):
                new[key] = spec.sample(rng)
            else:
                f = perturbation_factors[0] if rng.random() > 0.5 else perturbation_factors[1]
                v = cur * f
                if isinstance(spec, (Float, Integer)) and spec.lower is not None:
                    hi = spec.upper if isinstance(spec, Float) else spec.upper - 1
                    v = min(max(v, spec.lower), hi)
                new[key] = int(round(v)) if isinstance(spec, Integer) or isinstance(cur, int) and not isinstance(cur, bool) else float(v)
        elif callable(spec):
            new[key] = spec()
        else:
            raise TypeError(f"unsupported mutation spec for {key!r}: {spec!r}")
    if custom_explore_fn is not None:
        new = custom_explore_fn(new)
    return new


class PopulationBasedTraining(TrialScheduler):
    def __init__(self, time_attr: str = "training_iteration", metric: Optional[str] = None,
                 mode: Optional[str] = None, perturbation_interval: float = 4,
                 hyperparam_mutations: Optional[Dict[str, Any]] = None, quantile_fraction: float = 0.25,
                 resample_probability: float = 0.25, perturbation_factors: Tuple[float, float] = (1.2, 0.8),
                 custom_explore_fn: Optional[Callable[[Dict], Dict]] = None, seed: int = 0,
                 synch: bool = False):
        super().__init__(metric, mode)
        if not hyperparam_mutations and custom_explore_fn is None:
            raise ValueError("PBT needs hyperparam_mutations or a custom_explore_fn")
        if not 0 <= quantile_fraction <= 0.5:
            raise ValueError("quantile_fraction must be in [0, 0.5]")
        self.time_attr, self.interval = time_attr, perturbation_interval
        self.mutations = hyperparam_mutations or {}
        self.quantile_fraction, self.resample_probability = quantile_fraction, resample_probability
        self.perturbation_factors, self.custom_explore_fn = perturbation_factors, custom_explore_fn
        self._rng = np.random.default_rng(seed)
        self.synch = synch
        self._score: Dict[str, float] = {}
        self._last_perturb: Dict[str, float] = {}
        self.num_perturbations = 0
        self.exploit_log: List[Dict] = []      # [{"t":..,"trial":..,"donor":..,"old":..,"new":..}]

    # ------------------------------------------------------------------ helpers
    def _quantiles(self, ids: Sequence[str]) -> Tuple[List[str], List[str]]:
        scored = sorted((tid for tid in ids if tid in self._score), key=lambda t: (self._score[t], t))
        if len(scored) <= 1:
            return [], []
        n = int(np.ceil(len(scored) * self.quantile_fraction))
        n = min(n, len(scored) // 2)
        if n == 0:
            return [], []
        return scored[:n], scored[-n:]

    def _exploit(self, runner, trial: Trial, donor: Trial, t: float) -> None:
        new_cfg = explore(donor.config, self.mutations, self.resample_probability, self._rng,
                          self.perturbation_factors, self.custom_explore_fn)
        self.exploit_log.append({"t": t, "trial": trial.trial_id, "donor": donor.trial_id,
                                 "old": trial.config, "new": new_cfg})
        self.num_perturbations += 1
        trial.pbt_request = {"donor": donor.trial_id, "config": new_cfg}

    def decides_at(self, iteration: int) -> bool:
        """Members that report every iteration from 0 are perturbed at multiples of the interval and nowhere else."""
        if self.time_attr != "training_iteration" or float(self.interval) != int(self.interval) or self.interval < 1:
            return True
        return iteration % int(self.interval) == 0

    # ------------------------------------------------------------------ events
    def on_trial_result(self, runner, trial: Trial, result: Dict) -> str:
        t = result.get(self.time_attr, 0)
        v = self._signed(result.get(self.metric))
        self._score[trial.trial_id] = -np.inf if is_nan(v) else v
        last = self._last_perturb.get(trial.trial_id, 0)
        if t - last < self.interval:
            return CONTINUE
        self._last_perturb[trial.trial_id] = t
        live = [x.trial_id for x in runner.trials if not x.is_finished]
        lower, upper = self._quantiles(live)
        if trial.trial_id in lower and upper:
            donor_id = upper[int(self._rng.integers(len(upper)))]
            donor = next(x for x in runner.trials if x.trial_id == donor_id)
            if donor.checkpoint is None and not getattr(runner, "executor_moves_weights", False):
                return CONTINUE      # nothing to copy yet
            self._exploit(runner, trial, donor, t)
            return RESTART
        return CONTINUE

    def on_batch_result(self, runner, trials: Sequence[Trial], results: Sequence[Dict]) -> List[str]:
        """Lock-step population update: score everyone first, then exploit the bottom quantile."""
        order = sorted(range(len(trials)), key=lambda i: trials[i].trial_id)
        out = [CONTINUE] * len(trials)
        due: List[int] = []
        for i in order:
            t = results[i].get(self.time_attr, 0)
            v = self._signed(results[i].get(self.metric))
            self._score[trials[i].trial_id] = -np.inf if is_nan(v) else v
            if t - self._last_perturb.get(trials[i].trial_id, 0) >= self.interval:
                due.append(i)
        if not due:
            return out
        live = [x.trial_id for x in runner.trials if not x.is_finished]
        lower, upper = self._quantiles(live)
        by_id = {x.trial_id: x for x in runner.trials}
        for i in due:
            tr = trials[i]
            t = results[i].get(self.time_attr, 0)
            self._last_perturb[tr.trial_id] = t
            if tr.trial_id in lower and upper:
                donor = by_id[upper[int(self._rng.integers(len(upper)))]]
                self._exploit(runner, tr, donor, t)
                out[i] = RESTART
        return out

    def on_trial_remove(self, runner, trial: Trial) -> None:
        self._score.pop(trial.trial_id, None)

    on_trial_complete = lambda self, runner, trial, result: self._score.pop(trial.trial_id, None)  # noqa: E731

    def state_dict(self) -> Dict:
        return {"score": self._score, "last_perturb": self._last_perturb,
                "num_perturbations": self.num_perturbations,
                "rng": self._rng.bit_generator.state}

    def load_state_dict(self, state: Dict) -> None:
        self._score = dict(state.get("score", {}))
        self._last_perturb = dict(state.get("last_perturb", {}))
        self.num_perturbations = state.get("num_perturbations", 0)
        if "rng" in state:
            self._rng.bit_generator.state = state["rng"]

    def debug_string(self) -> str:
        return f"PBT: {self.num_perturbations} perturbations"
