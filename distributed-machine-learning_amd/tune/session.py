"""Per-trial session: the trainable's side of the driver <-> trial channel.

``report(**metrics)`` is the function-API reporter the reference calls once per trial
(/root/reference/ray-tune-hpo-regression.py:373,
/root/reference/ray-tune-hpo-regression-sample.py:135).  It hands the metrics to the driver,
blocks until the scheduler's decision comes back and unwinds the trainable when that decision
is STOP.  Reporting once per epoch (instead of once per trial) is what lets ASHA prune.
"""
from __future__ import annotations

import os
import socket
import threading
import time
from typing import Any, Callable, Dict, Optional

CONTINUE = "CONTINUE"
STOP = "STOP"
PAUSE = "PAUSE"
RESTART = "RESTART"   # stop now; the driver relaunches the trial with a new config/checkpoint


class StopTrial(BaseException):
    """Raised inside the trainable by ``report`` when the scheduler stops the trial.

    Derives from ``BaseException`` so a user's ``except Exception`` does not swallow it.
    """

    def __init__(self, decision: str = STOP):
        super().__init__(decision)
        self.decision = decision


class Session:
    """State shared between one running trainable and its executor."""

    def __init__(self, trial_id: str, config: Dict, send: Callable[[Dict], str],
                 checkpoint: Any = None, logdir: Optional[str] = None, experiment_id: str = "",
                 iteration_offset: int = 0, time_offset: float = 0.0):
        self.trial_id = trial_id
        self.config = config
        self._send = send
        self.checkpoint = checkpoint
        self.logdir = logdir
        self.experiment_id = experiment_id
        self.iteration = iteration_offset
        self.iterations_since_restore = 0
        self._t_start = time.time()
        self._t_last = self._t_start
        self._time_offset = time_offset

    def report(self, metrics: Dict, checkpoint: Any = None) -> None:
        now = time.time()
        self.iteration += 1
        self.iterations_since_restore += 1
        result = dict(metrics)
        result.setdefault("training_iteration", self.iteration)
        result.update(
            time_this_iter_s=now - self._t_last,
            time_total_s=self._time_offset + (now - self._t_start),
            timestamp=int(now),
            date=time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime(now)),
            done=False,
            trial_id=self.trial_id,
            experiment_id=self.experiment_id,
            pid=os.getpid(),
            hostname=socket.gethostname(),
            node_ip="127.0.0.1",
            iterations_since_restore=self.iterations_since_restore,
        )
        self._t_last = now
        decision = self._send({"type": "result", "trial_id": self.trial_id, "result": result,
                               "checkpoint": checkpoint})
        if decision in (STOP, PAUSE, RESTART):
            raise StopTrial(decision)


_local = threading.local()
_process_session: Optional[Session] = None


def _set_session(s: Optional[Session], process_wide: bool = False) -> None:
    global _process_session
    if process_wide:
        _process_session = s
    else:
        _local.session = s


def get_session() -> Optional[Session]:
    s = getattr(_local, "session", None)
    return s if s is not None else _process_session


def report(_metrics: Optional[Dict] = None, *, checkpoint: Any = None, **kwargs) -> None:
    """Report metrics for the current training iteration.

    Accepts both the keyword style the reference uses (``report(validation_loss=x)``) and a
    single dict.  ``checkpoint`` may be any picklable object (typically a dict holding
    ``state_dict``s); the driver keeps the newest one per trial and persists it under
    ``checkpoint_<iter>/``.
    """
    metrics = dict(_metrics or {})
    metrics.update(kwargs)
    s = get_session()
    if s is None:
        raise RuntimeError("mitune.report() called outside of a trial started by mitune.run()")
    s.report(metrics, checkpoint=checkpoint)


def get_checkpoint() -> Any:
    """Checkpoint to resume from (after a failure retry, a PBT exploit or an experiment resume)."""
    s = get_session()
    return None if s is None else s.checkpoint


def get_trial_id() -> Optional[str]:
    s = get_session()
    return None if s is None else s.trial_id


def get_trial_dir() -> Optional[str]:
    s = get_session()
    return None if s is None else s.logdir


def is_session_enabled() -> bool:
    return get_session() is not None
