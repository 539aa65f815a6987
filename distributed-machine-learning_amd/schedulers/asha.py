"""Asynchronous Successive Halving (ASHA) and its rung-synchronous variant.

Behavioural target: the scheduler the reference constructs as
``ASHAScheduler(metric=..., mode="min")`` (/root/reference/ray-tune-hpo-regression.py:473,
/root/reference/ray-tune-hpo-regression-sample.py:163) with its defaults
``time_attr="training_iteration", max_t=100, grace_period=1, reduction_factor=4, brackets=1``:
rungs at ``grace_period * rf**k`` below ``max_t`` (1, 4, 16, 64); at the first result past a rung
a trial is stopped when its sign-normalised metric is below the ``(1 - 1/rf)`` percentile of the
values recorded at that rung so far.

``on_batch_result`` is the collective form used when many trials report the same iteration at
once (a GPU pack, or metrics all-gathered from every rank): all values are recorded first and
then cut at the percentile, which is successive halving proper and -- unlike the one-at-a-time
rule -- independent of arrival order, so replicated schedulers on every rank agree bit-for-bit.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from .base import TrialScheduler, CONTINUE, STOP, is_nan
from ..tune.trial import Trial


def _percentile_cutoff(values: Sequence[float], rf: float) -> Optional[float]:
    vals = [v for v in values if not is_nan(v)]
    if not vals:
        return None
    return float(np.nanpercentile(np.asarray(vals, dtype=np.float64), (1.0 - 1.0 / rf) * 100.0))


class _Bracket:
    def __init__(self, min_t: float, max_t: float, rf: float, s: int):
        self.rf = rf
        n_rungs = int(math.log(max_t / min_t) / math.log(rf) - s + 1)
        # highest rung first, so a late first report lands on the deepest rung it qualifies for
        self.rungs: List[Dict] = [
            {"milestone": min_t * rf ** (k + s), "recorded": {}} for k in reversed(range(max(n_rungs, 0)))
        ]

    def on_result(self, trial_id: str, t: float, value: Optional[float]) -> str:
        action = CONTINUE
        for rung in self.rungs:
            if t < rung["milestone"] or trial_id in rung["recorded"]:
                continue
            cutoff = _percentile_cutoff(list(rung["recorded"].values()), self.rf)
            if is_nan(value):
                action = STOP       # a diverged trial never survives a rung
            else:
                if cutoff is not None and value < cutoff:
                    action = STOP
                rung["recorded"][trial_id] = value
            break
        return action

    def on_batch(self, ids: Sequence[str], ts: Sequence[float], values: Sequence[Optional[float]]) -> List[str]:
        """Record every value at its rung, then cut each touched rung once."""
        actions = [CONTINUE] * len(ids)
        touched: Dict[int, List[int]] = {}
        for i, (tid, t, v) in enumerate(zip(ids, ts, values)):
            for r, rung in enumerate(self.rungs):
                if t < rung["milestone"] or tid in rung["recorded"]:
                    continue
                if is_nan(v):
                    actions[i] = STOP
                else:
                    rung["recorded"][tid] = v
                    touched.setdefault(r, []).append(i)
                break
        for r, members in touched.items():
            rung = self.rungs[r]
            cutoff = _percentile_cutoff(list(rung["recorded"].values()), self.rf)
            if cutoff is None:
                continue
            for i in members:
                if values[i] < cutoff:
                    actions[i] = STOP
        return actions

    def state(self) -> List[Dict]:
        return [{"milestone": r["milestone"], "recorded": dict(r["recorded"])} for r in self.rungs]

    def debug(self) -> str:
        return " | ".join(f"t={r['milestone']:g}: n={len(r['recorded'])}" for r in reversed(self.rungs))


class AsyncHyperBandScheduler(TrialScheduler):
    def __init__(self, time_attr: str = "training_iteration", metric: Optional[str] = None,
                 mode: Optional[str] = None, max_t: float = 100, grace_period: float = 1,
                 reduction_factor: float = 4, brackets: int = 1, seed: int = 0):
        super().__init__(metric, mode)
        if max_t <= 0 or grace_period <= 0 or max_t < grace_period:
            raise ValueError("need 0 < grace_period <= max_t")
        if reduction_factor <= 1:
            raise ValueError("reduction_factor must be > 1")
        self.time_attr, self.max_t, self.grace_period = time_attr, max_t, grace_period
        self.rf, self.num_brackets = reduction_factor, brackets
        self._brackets = [_Bracket(grace_period, max_t, reduction_factor, s) for s in range(brackets)]
        self._trial_bracket: Dict[str, int] = {}
        self._rng = np.random.default_rng(seed)
        self.num_stopped = 0

    # rung milestones of bracket 0, ascending -- handy for executors that align epochs to rungs
    @property
    def milestones(self) -> List[float]:
        return sorted(r["milestone"] for r in self._brackets[0].rungs)

    def decides_at(self, iteration: int) -> bool:
        """A trial is only ever cut at a rung it has just reached (or at ``max_t``): between rungs its results are recorded
        nowhere and answered CONTINUE.  Exact for trials that report every iteration from their first one."""
        if self.time_attr != "training_iteration":
            return True
        return iteration >= self.max_t or any(abs(iteration - r["milestone"]) < 1e-9 for b in self._brackets for r in b.rungs)

    def on_trial_add(self, runner, trial: Trial) -> None:
        if trial.trial_id in self._trial_bracket:
            return
        if self.num_brackets == 1:
            idx = 0
        else:   # softmax over bracket sizes, as in the HyperBand paper's sampling of s
            sizes = np.array([len(b.rungs) for b in self._brackets], dtype=np.float64)
            p = np.exp(sizes - sizes.max())
            idx = int(self._rng.choice(len(self._brackets), p=p / p.sum()))
        self._trial_bracket[trial.trial_id] = idx

    def _value(self, result: Dict) -> Optional[float]:
        if self.metric is None:
            raise ValueError("ASHA needs a metric: pass metric/mode to the scheduler or to run()")
        return self._signed(result.get(self.metric))

    def on_trial_result(self, runner, trial: Trial, result: Dict) -> str:
        if trial.trial_id not in self._trial_bracket:
            self.on_trial_add(runner, trial)
        t = result.get(self.time_attr, 0)
        if t >= self.max_t:
            action = STOP
            # still record, so peers are compared against it
            self._brackets[self._trial_bracket[trial.trial_id]].on_result(trial.trial_id, t, self._value(result))
        else:
            action = self._brackets[self._trial_bracket[trial.trial_id]].on_result(
                trial.trial_id, t, self._value(result))
        if action == STOP:
            self.num_stopped += 1
        return action

    def on_batch_result(self, runner, trials: Sequence[Trial], results: Sequence[Dict]) -> List[str]:
        order = sorted(range(len(trials)), key=lambda i: trials[i].trial_id)
        out = [CONTINUE] * len(trials)
        by_bracket: Dict[int, List[int]] = {}
        for i in order:
            if trials[i].trial_id not in self._trial_bracket:
                self.on_trial_add(runner, trials[i])
            by_bracket.setdefault(self._trial_bracket[trials[i].trial_id], []).append(i)
        for b, members in sorted(by_bracket.items()):
            acts = self._brackets[b].on_batch(
                [trials[i].trial_id for i in members],
                [results[i].get(self.time_attr, 0) for i in members],
                [self._value(results[i]) for i in members])
            for i, a in zip(members, acts):
                if results[i].get(self.time_attr, 0) >= self.max_t:
                    a = STOP
                out[i] = a
        self.num_stopped += sum(a == STOP for a in out)
        return out

    def on_trial_complete(self, runner, trial: Trial, result: Dict) -> None:
        if result and self.metric in result and trial.trial_id in self._trial_bracket:
            self._brackets[self._trial_bracket[trial.trial_id]].on_result(
                trial.trial_id, result.get(self.time_attr, 0), self._value(result))

    def on_trial_remove(self, runner, trial: Trial) -> None:
        self._trial_bracket.pop(trial.trial_id, None)

    def state_dict(self) -> Dict:
        return {"brackets": [b.state() for b in self._brackets],
                "trial_bracket": dict(self._trial_bracket), "num_stopped": self.num_stopped}

    def load_state_dict(self, state: Dict) -> None:
        for b, rungs in zip(self._brackets, state.get("brackets", [])):
            for rung, saved in zip(b.rungs, rungs):
                rung["recorded"] = dict(saved["recorded"])
        self._trial_bracket = dict(state.get("trial_bracket", {}))
        self.num_stopped = state.get("num_stopped", 0)

    def debug_string(self) -> str:
        return "ASHA " + " || ".join(f"bracket {i}: {b.debug()}" for i, b in enumerate(self._brackets))


ASHAScheduler = AsyncHyperBandScheduler


class MedianStoppingRule(TrialScheduler):
    """Stop a trial whose best result so far is worse than the median of its peers' running means."""

    def __init__(self, time_attr: str = "training_iteration", metric: Optional[str] = None,
                 mode: Optional[str] = None, grace_period: float = 1, min_samples_required: int = 3):
        super().__init__(metric, mode)
        self.time_attr, self.grace_period, self.min_samples = time_attr, grace_period, min_samples_required
        self._history: Dict[str, List[float]] = {}

    def on_trial_result(self, runner, trial: Trial, result: Dict) -> str:
        v = self._signed(result.get(self.metric))
        if is_nan(v):
            return STOP
        hist = self._history.setdefault(trial.trial_id, [])
        hist.append(v)
        t = result.get(self.time_attr, 0)
        if t < self.grace_period:
            return CONTINUE
        means = [float(np.mean(h[:len(hist)])) for tid, h in self._history.items()
                 if tid != trial.trial_id and len(h) > 0]
        if len(means) < self.min_samples:
            return CONTINUE
        return STOP if max(hist) < float(np.median(means)) else CONTINUE

    def state_dict(self) -> Dict:
        return {"history": self._history}

    def load_state_dict(self, state: Dict) -> None:
        self._history = {k: list(v) for k, v in state.get("history", {}).items()}
