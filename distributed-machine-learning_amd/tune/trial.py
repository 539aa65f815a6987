"""Trial record and life-cycle states.

Mirrors the trial life-cycle the reference relies on implicitly through ``tune.run``
(/root/reference/ray-tune-hpo-regression.py:469-478): PENDING -> RUNNING ->
{TERMINATED, ERROR}, plus PAUSED for schedulers that park trials (PBT, synchronous rungs).
"""
from __future__ import annotations

import json
import os
import re
import time
import uuid
from typing import Any, Dict, List, Optional

from .space import flatten

PENDING = "PENDING"
RUNNING = "RUNNING"
PAUSED = "PAUSED"
TERMINATED = "TERMINATED"
ERROR = "ERROR"

_ALL_STATES = (PENDING, RUNNING, PAUSED, TERMINATED, ERROR)


def _short(v: Any) -> str:
    if isinstance(v, float):
        return f"{v:.4g}"
    return str(v)


def make_trial_dirname(trainable_name: str, trial_id: str, index: int, config: Dict, limit: int = 96) -> str:
    """``<fn>_<trialid>_<idx>_<k=v,...>`` shortened to a filesystem-safe name."""
    kv = ",".join(f"{k.split('/')[-1]}={_short(v)}" for k, v in sorted(flatten(config).items())
                  if isinstance(v, (int, float, str, bool)))
    name = f"{trainable_name}_{trial_id}_{index:05d}_{kv}"
    name = re.sub(r"[^A-Za-z0-9_.,=+-]", "_", name)
    return name[:limit].rstrip(",_")


class Trial:
    """One hyper-parameter configuration and everything the driver knows about its run."""

    def __init__(self, config: Dict, trial_id: Optional[str] = None, index: int = 0,
                 trainable_name: str = "trainable", experiment_dir: Optional[str] = None,
                 experiment_tag: str = ""):
        self.trial_id = trial_id or uuid.uuid4().hex[:8]
        self.index = index
        self.config = config
        self.trainable_name = trainable_name
        self.status = PENDING
        self.results: List[Dict] = []          # every reported result, in order
        self.last_result: Dict = {}
        self.metric_best: Dict[str, Dict[str, float]] = {}   # metric -> {"min":..,"max":..}
        self.error_msg: Optional[str] = None
        self.num_failures = 0
        self.start_time: Optional[float] = None
        self.end_time: Optional[float] = None
        self.checkpoint: Any = None             # newest in-memory checkpoint object (or path)
        self.checkpoint_iter: int = 0
        self.restore_from: Any = None           # checkpoint to hand to the trainable at (re)start
        self.logdir: Optional[str] = None
        self.experiment_tag = experiment_tag
        self.placement: Dict[str, Any] = {}     # e.g. {"gpu": 3, "pack": 7, "slot": 12}
        if experiment_dir is not None:
            self.logdir = os.path.join(
                experiment_dir, make_trial_dirname(trainable_name, self.trial_id, index, config))

    # ------------------------------------------------------------------ properties
    @property
    def training_iteration(self) -> int:
        return int(self.last_result.get("training_iteration", 0))

    @property
    def is_finished(self) -> bool:
        return self.status in (TERMINATED, ERROR)

    def set_status(self, status: str) -> None:
        assert status in _ALL_STATES, status
        if status == RUNNING and self.start_time is None:
            self.start_time = time.time()
        if status in (TERMINATED, ERROR):
            self.end_time = time.time()
        self.status = status

    def record_result(self, result: Dict) -> None:
        self.results.append(result)
        self.last_result = result
        best = self.metric_best
        for k, v in result.items():
            tv = type(v)
            if tv is float or tv is int:
                b = best.get(k)
                if b is None:
                    best[k] = {"min": v, "max": v}
                elif v == v:  # NaN never becomes a best value
                    lo, hi = b["min"], b["max"]
                    if lo != lo or v < lo:
                        b["min"] = v
                    if hi != hi or v > hi:
                        b["max"] = v

    # ------------------------------------------------------------------ persistence
    def to_json(self) -> Dict:
        return {
            "trial_id": self.trial_id, "index": self.index, "config": self.config,
            "trainable_name": self.trainable_name, "status": self.status,
            "last_result": self.last_result, "num_results": len(self.results),
            "metric_best": self.metric_best, "error_msg": self.error_msg,
            "num_failures": self.num_failures, "start_time": self.start_time,
            "end_time": self.end_time, "logdir": self.logdir,
            "checkpoint_iter": self.checkpoint_iter, "placement": self.placement,
        }

    @classmethod
    def from_json(cls, d: Dict) -> "Trial":
        t = cls(d["config"], trial_id=d["trial_id"], index=d.get("index", 0),
                trainable_name=d.get("trainable_name", "trainable"))
        t.status = d["status"]
        t.last_result = d.get("last_result") or {}
        t.metric_best = d.get("metric_best") or {}
        t.error_msg = d.get("error_msg")
        t.num_failures = d.get("num_failures", 0)
        t.start_time, t.end_time = d.get("start_time"), d.get("end_time")
        t.logdir = d.get("logdir")
        t.checkpoint_iter = d.get("checkpoint_iter", 0)
        t.placement = d.get("placement") or {}
        # the result history lives in result.json next to the trial, reload it lazily
        if t.logdir and os.path.exists(os.path.join(t.logdir, "result.json")):
            with open(os.path.join(t.logdir, "result.json")) as f:
                t.results = [json.loads(line) for line in f if line.strip()]
        elif t.last_result:
            t.results = [t.last_result]
        return t

    def __repr__(self):
        return f"Trial({self.trial_id}, {self.status}, it={self.training_iteration})"
