"""Per-trial result loggers and the experiment directory layout.

Layout written under ``local_dir`` (the reference only names the directory,
/root/reference/ray-tune-hpo-regression.py:476; the tree below is what a ``tune.run`` user
expects to find there):

    <local_dir>/<experiment_name>/
        experiment_state.json              trial table + scheduler/searcher state (resume)
        <trainable>_<trial_id>_<idx>_<k=v>/
            params.json                    resolved config
            result.json                    JSON lines, one per report
            progress.csv                   same, tabular
            events.out.tfevents.<ts>.<host>   same, as TensorBoard scalars (tune/tbevents.py)
            params.pkl                     pickled config (keeps tuples / numpy scalars exactly)
            error.txt                      traceback if the trial failed
            checkpoint_<iter:06d>/         model.pt / optimizer.pt / meta.json
"""
from __future__ import annotations

import csv
import json
import math
import os
from typing import Any, Dict, List, Optional

from .space import flatten
from .trial import Trial


def _jsonable(o: Any) -> Any:
    try:
        import numpy as np
        if isinstance(o, np.generic):
            o = o.item()
        elif isinstance(o, np.ndarray):
            return o.tolist()
    except ImportError:  # pragma: no cover
        pass
    try:
        import torch
        if isinstance(o, torch.Tensor):
            return o.item() if o.numel() == 1 else o.tolist()
    except ImportError:  # pragma: no cover
        pass
    if isinstance(o, float):
        if math.isnan(o):
            return "NaN"
        if math.isinf(o):
            return "Infinity" if o > 0 else "-Infinity"
        return o
    if isinstance(o, (str, int, bool)) or o is None:
        return o
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple, set)):
        return [_jsonable(v) for v in o]
    return repr(o)


def dump_json(obj: Any, path: str) -> None:
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(_jsonable(obj), f, indent=1, sort_keys=True)
    os.replace(tmp, path)     # atomic: a crash never leaves a half-written state file


class Logger:
    def on_trial_start(self, trial: Trial) -> None:
        pass

    def on_result(self, trial: Trial, result: Dict) -> None:
        pass

    def on_trial_end(self, trial: Trial) -> None:
        pass

    def close(self) -> None:
        pass


class JsonLogger(Logger):
    def on_trial_start(self, trial):
        if trial.logdir:
            os.makedirs(trial.logdir, exist_ok=True)
            dump_json(trial.config, os.path.join(trial.logdir, "params.json"))
            try:       # the pickled twin keeps non-JSON values (tuples, numpy scalars) exactly
                import pickle
                with open(os.path.join(trial.logdir, "params.pkl"), "wb") as f:
                    pickle.dump(trial.config, f)
            except Exception:       # noqa: BLE001 - an unpicklable config value must not fail the trial
                pass

    def on_result(self, trial, result):
        if trial.logdir:
            with open(os.path.join(trial.logdir, "result.json"), "a") as f:
                f.write(json.dumps(_jsonable(result), sort_keys=True) + "\n")

    def on_trial_end(self, trial):
        if trial.logdir and trial.error_msg:
            with open(os.path.join(trial.logdir, "error.txt"), "w") as f:
                f.write(trial.error_msg)


class CSVLogger(Logger):
    def __init__(self):
        self._files: Dict[str, Any] = {}
        self._writers: Dict[str, csv.DictWriter] = {}

    def on_result(self, trial, result):
        if not trial.logdir:
            return
        flat = {k: v for k, v in flatten(result).items() if not isinstance(v, (dict, list))}
        w = self._writers.get(trial.trial_id)
        if w is None:
            os.makedirs(trial.logdir, exist_ok=True)
            path = os.path.join(trial.logdir, "progress.csv")
            new = not os.path.exists(path) or os.path.getsize(path) == 0
            f = open(path, "a", newline="")
            w = csv.DictWriter(f, fieldnames=list(flat.keys()), extrasaction="ignore")
            if new:
                w.writeheader()
            self._files[trial.trial_id], self._writers[trial.trial_id] = f, w
        w.writerow(flat)
        self._files[trial.trial_id].flush()

    def on_trial_end(self, trial):
        f = self._files.pop(trial.trial_id, None)
        self._writers.pop(trial.trial_id, None)
        if f:
            f.close()

    def close(self):
        for f in self._files.values():
            f.close()
        self._files.clear()
        self._writers.clear()


class MemoryLogger(Logger):
    """Keeps everything in lists; used by tests and by ranks that must not touch the disk."""

    def __init__(self):
        self.rows: List[Dict] = []

    def on_result(self, trial, result):
        self.rows.append({"trial_id": trial.trial_id, **result})


DEFAULT_LOGGERS = (JsonLogger, CSVLogger)
