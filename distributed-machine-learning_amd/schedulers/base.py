"""Scheduler interface and the FIFO scheduler.

A scheduler sees every reported result and answers CONTINUE / PAUSE / STOP.  The batch entry
point ``on_batch_result`` exists for executors that advance many trials in lock-step (a GPU
pack, or the all-gathered metrics of an 8-GPU job): it must be a pure function of the scheduler
state and the (trial-id sorted) batch so that every rank reaches identical decisions.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

from ..tune.session import CONTINUE, PAUSE, STOP, RESTART  # noqa: F401  (re-exported)
from ..tune.trial import Trial, PENDING, PAUSED


class TrialScheduler:
    CONTINUE, PAUSE, STOP, RESTART = CONTINUE, PAUSE, STOP, RESTART

    def __init__(self, metric: Optional[str] = None, mode: Optional[str] = None):
        if mode not in (None, "min", "max"):
            raise ValueError("mode must be 'min' or 'max'")
        self.metric, self.mode = metric, mode

    # ------------------------------------------------------------------ configuration
    def set_search_properties(self, metric: Optional[str], mode: Optional[str]) -> bool:
        """Adopt the experiment's metric/mode when the scheduler was built without one."""
        if self.metric and metric and self.metric != metric:
            return False
        if self.mode and mode and self.mode != mode:
            return False
        self.metric = self.metric or metric
        self.mode = self.mode or mode
        return True

    def _signed(self, value: Optional[float]) -> Optional[float]:
        """Metric normalised so that larger is always better (mode=min -> negate)."""
        if value is None:
            return None
        return -float(value) if self.mode == "min" else float(value)

    # ------------------------------------------------------------------ events
    def on_trial_add(self, runner, trial: Trial) -> None:
        pass

    def on_trial_result(self, runner, trial: Trial, result: Dict) -> str:
        return CONTINUE

    def on_batch_result(self, runner, trials: Sequence[Trial], results: Sequence[Dict]) -> List[str]:
        order = sorted(range(len(trials)), key=lambda i: trials[i].trial_id)
        out = [CONTINUE] * len(trials)
        for i in order:
            out[i] = self.on_trial_result(runner, trials[i], results[i])
        return out

    def decides_at(self, iteration: int) -> bool:
        """Can a result reported at this ``training_iteration`` be answered with anything but CONTINUE?  Executors that
        exchange results between processes use it to skip the wait for each other on rounds where nothing can be
        decided.  A scheduler that cannot tell says True."""
        return True

    def on_trial_complete(self, runner, trial: Trial, result: Dict) -> None:
        pass

    def on_trial_error(self, runner, trial: Trial) -> None:
        pass

    def on_trial_remove(self, runner, trial: Trial) -> None:
        pass

    def choose_trial_to_run(self, runner) -> Optional[Trial]:
        nxt = getattr(runner, "next_pending", None)
        if nxt is not None:                  # O(1): the runner keeps its PENDING trials in a queue
            t = nxt()
            if t is not None:
                return t
            if not runner.num_paused:
                return None
        else:
            for t in runner.trials:
                if t.status == PENDING:
                    return t
        for t in runner.trials:
            if t.status == PAUSED and getattr(t, "resumable", True):
                return t
        return None

    # ------------------------------------------------------------------ persistence
    def state_dict(self) -> Dict:
        return {}

    def load_state_dict(self, state: Dict) -> None:
        pass

    def debug_string(self) -> str:
        return type(self).__name__


class FIFOScheduler(TrialScheduler):
    """Run trials in submission order, never stop early."""

    def decides_at(self, iteration: int) -> bool:
        return False


def is_nan(x) -> bool:
    return x is None or (isinstance(x, float) and math.isnan(x))
