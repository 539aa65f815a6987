"""ExperimentAnalysis: best-trial queries over a finished (or reloaded) experiment.

The reference ends both scripts with ``analysis.best_config``
(/root/reference/ray-tune-hpo-regression.py:480,
/root/reference/ray-tune-hpo-regression-sample.py:170) after passing metric/mode to the
scheduler only.  Here the experiment inherits metric/mode from the scheduler or searcher when
``run()`` was given none, so that property works as the scripts intend.
"""
from __future__ import annotations

import json
import math
import os
from typing import Any, Dict, List, Optional

from .space import flatten
from .trial import Trial, ERROR


class ExperimentAnalysis:
    def __init__(self, trials: List[Trial], default_metric: Optional[str] = None,
                 default_mode: Optional[str] = None, experiment_dir: Optional[str] = None,
                 stats: Optional[Dict] = None):
        self.trials = trials
        self.default_metric, self.default_mode = default_metric, default_mode
        self.experiment_dir = experiment_dir
        self.stats = stats or {}

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_directory(cls, experiment_dir: str, default_metric: Optional[str] = None,
                       default_mode: Optional[str] = None) -> "ExperimentAnalysis":
        with open(os.path.join(experiment_dir, "experiment_state.json")) as f:
            state = json.load(f)
        trials = [Trial.from_json(t) for t in state["trials"]]
        return cls(trials, default_metric or state.get("metric"), default_mode or state.get("mode"),
                   experiment_dir, state.get("stats"))

    # ------------------------------------------------------------------ helpers
    def _mm(self, metric: Optional[str], mode: Optional[str]):
        metric = metric or self.default_metric
        mode = mode or self.default_mode
        if not metric or mode not in ("min", "max"):
            raise ValueError("No metric/mode known for this experiment: pass metric= and mode= to run(), "
                             "to the scheduler or the searcher, or to this call.")
        return metric, mode

    @staticmethod
    def _trial_score(trial: Trial, metric: str, mode: str, scope: str) -> Optional[float]:
        if scope == "last":
            v = trial.last_result.get(metric)
        elif scope == "all":
            v = trial.metric_best.get(metric, {}).get(mode)
        elif scope == "avg":
            vals = [r[metric] for r in trial.results if metric in r]
            v = sum(vals) / len(vals) if vals else None
        elif scope.startswith("last-") and scope.endswith("-avg"):
            n = int(scope.split("-")[1])
            vals = [r[metric] for r in trial.results if metric in r][-n:]
            v = sum(vals) / len(vals) if vals else None
        else:
            raise ValueError(f"unknown scope {scope!r}")
        if isinstance(v, str):          # "NaN" after a JSON round trip
            try:
                v = float(v)
            except ValueError:
                return None
        if v is None or (isinstance(v, float) and math.isnan(v)):
            return None
        return float(v)

    def get_best_trial(self, metric: Optional[str] = None, mode: Optional[str] = None,
                       scope: str = "last", filter_nan_and_inf: bool = True) -> Optional[Trial]:
        metric, mode = self._mm(metric, mode)
        best, best_v = None, None
        for t in self.trials:
            v = self._trial_score(t, metric, mode, scope)
            if v is None or (filter_nan_and_inf and math.isinf(v)):
                continue
            if best_v is None or (v < best_v if mode == "min" else v > best_v):
                best, best_v = t, v
        return best

    def get_best_config(self, metric=None, mode=None, scope="last") -> Optional[Dict]:
        t = self.get_best_trial(metric, mode, scope)
        return None if t is None else t.config

    def get_best_logdir(self, metric=None, mode=None, scope="last") -> Optional[str]:
        t = self.get_best_trial(metric, mode, scope)
        return None if t is None else t.logdir

    def get_best_checkpoint(self, trial: Optional[Trial] = None, metric=None, mode=None) -> Optional[str]:
        from .checkpoint import latest_checkpoint
        trial = trial or self.get_best_trial(metric, mode)
        return None if trial is None or not trial.logdir else latest_checkpoint(trial.logdir)

    # ------------------------------------------------------------------ properties (experiment defaults)
    @property
    def best_trial(self) -> Optional[Trial]:
        return self.get_best_trial()

    @property
    def best_config(self) -> Optional[Dict]:
        return self.get_best_config()

    @property
    def best_result(self) -> Optional[Dict]:
        t = self.best_trial
        return None if t is None else t.last_result

    @property
    def best_logdir(self) -> Optional[str]:
        return self.get_best_logdir()

    @property
    def best_checkpoint(self) -> Optional[str]:
        return self.get_best_checkpoint()

    @property
    def results(self) -> Dict[str, Dict]:
        return {t.trial_id: t.last_result for t in self.trials}

    @property
    def errored_trials(self) -> List[Trial]:
        return [t for t in self.trials if t.status == ERROR]

    @property
    def trial_dataframes(self):
        import pandas as pd
        return {t.trial_id: pd.DataFrame([flatten(r) for r in t.results]) for t in self.trials}

    @property
    def results_df(self):
        return self.dataframe()

    def dataframe(self, metric: Optional[str] = None, mode: Optional[str] = None):
        """One row per trial: last result (or the best row by ``metric``) + ``config/<key>`` columns."""
        import pandas as pd
        rows = []
        for t in self.trials:
            row: Dict[str, Any] = {}
            if metric and mode and t.results:
                cand = [r for r in t.results if metric in r and r[metric] == r[metric]]
                if cand:
                    row.update(flatten((min if mode == "min" else max)(cand, key=lambda r: r[metric])))
            else:
                row.update(flatten(t.last_result))
            row.update({f"config/{k}": v for k, v in flatten(t.config).items()})
            row["trial_id"], row["status"], row["logdir"] = t.trial_id, t.status, t.logdir
            rows.append(row)
        return pd.DataFrame(rows)
