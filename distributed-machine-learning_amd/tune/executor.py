"""Trial executors for function trainables.

The reference delegates trial placement to Ray: one actor process per trial with
``resources_per_trial={"cpu": 2, "gpu": 1}`` and the trial's GPU exposed as ``cuda:0``
(/root/reference/ray-tune-hpo-regression.py:475,286;
/root/reference/ray-tune-hpo-regression-sample.py:164,97).  Here the same contract is served
by two light executors that need no cluster runtime:

* ``ThreadExecutor``  - each trial runs in a thread of the driver process.  Zero-copy sharing of
  ``with_parameters`` objects (datasets) and the right choice for CPU plumbing / tests.
* ``ProcessExecutor`` - each trial runs in a spawned worker process whose ``HIP_VISIBLE_DEVICES``
  is narrowed to its own GPU(s), so ``cuda:0`` inside the trainable is the trial's GPU.

Packed GPU execution (many trials advanced by one kernel launch) is a third executor living in
``mitune.packed.executor``; all three speak the event protocol below to the ``TrialRunner``.
"""
from __future__ import annotations

import multiprocessing as mp
import os
import contextlib
import queue
import threading
import traceback
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional

from . import session as _session
from .trial import Trial


@dataclass
class Event:
    kind: str                 # "result" | "done" | "error"
    trial: Trial
    result: Optional[Dict] = None
    checkpoint: Any = None
    error: Optional[str] = None


class Executor:
    """Interface the runner drives."""

    def has_capacity(self) -> bool:
        raise NotImplementedError

    def start_trial(self, trial: Trial) -> None:
        raise NotImplementedError

    def get_events(self, timeout: Optional[float] = None) -> List[Event]:
        raise NotImplementedError

    def resolve(self, trial: Trial, decision: str) -> None:
        """Answer the ``result`` event of ``trial`` with a scheduler decision."""
        raise NotImplementedError

    def num_running(self) -> int:
        raise NotImplementedError

    def shutdown(self) -> None:
        pass


def _resource_slots(resources_per_trial: Optional[Dict], max_concurrent: Optional[int]) -> tuple:
    """How many trials may run at once, and the GPU ids that can be handed out."""
    res = dict(resources_per_trial or {})
    cpus_per = float(res.get("cpu", 1) or 1)
    gpus_per = float(res.get("gpu", 0) or 0)
    n_cpu = os.cpu_count() or 1
    slots = max(1, int(n_cpu // max(cpus_per, 1e-9)))
    gpu_ids: List[int] = []
    if gpus_per > 0:
        try:
            import torch
            n_gpu = torch.cuda.device_count()
        except Exception:  # pragma: no cover
            n_gpu = 0
        if n_gpu > 0:
            gpu_ids = list(range(n_gpu))
            slots = min(slots, max(1, int(n_gpu // gpus_per)))
        # With no GPU present Ray would leave GPU trials pending forever; running them on the
        # CPU keeps scripts written for a GPU box usable on a laptop or in CI.
    if max_concurrent:
        slots = min(slots, max_concurrent)
    return slots, gpu_ids, gpus_per


def physical_device_ids(logical: List[int], env: Optional[Dict[str, str]] = None) -> List[str]:
    """Entries of the parent's own device mask for the logical device indices ``logical``.

    A process started under ``HIP_VISIBLE_DEVICES=4,5`` sees its devices as 0 and 1; a child that should get "device 1"
    must be started with ``HIP_VISIBLE_DEVICES=5``, not ``1`` -- index 1 of the whole machine belongs to somebody
    else.  With no mask in force the mapping is the identity.  Entries may be indices or device UUIDs.
    """
    env = os.environ if env is None else env
    mask = env.get("HIP_VISIBLE_DEVICES")
    if mask is None or mask.strip() == "":
        mask = env.get("CUDA_VISIBLE_DEVICES")
    if mask is None or mask.strip() == "":
        return [str(i) for i in logical]
    entries = [e.strip() for e in mask.split(",") if e.strip() != ""]
    out = []
    for i in logical:
        if not 0 <= i < len(entries):
            raise ValueError(f"logical device {i} is outside this process's device mask {mask!r}")
        out.append(entries[i])
    return out


class ThreadExecutor(Executor):
    def __init__(self, trainable: Callable, resources_per_trial: Optional[Dict] = None,
                 max_concurrent: Optional[int] = None, experiment_id: str = ""):
        self.trainable = trainable
        self.slots, self.gpu_ids, self.gpus_per = _resource_slots(resources_per_trial, max_concurrent)
        self.experiment_id = experiment_id
        self._events: "queue.Queue[Event]" = queue.Queue()
        self._replies: Dict[str, "queue.Queue[str]"] = {}
        self._threads: Dict[str, threading.Thread] = {}
        self._lock = threading.Lock()
        self._shared_flag_before: Optional[bool] = None     # train.graphed.SHARED_GPU_THREADS as found, once changed

    def has_capacity(self) -> bool:
        return len(self._threads) < self.slots

    def num_running(self) -> int:
        return len(self._threads)

    def _stream_scope(self):
        """Trials that share a GPU (fractional ``gpu``) each run on a HIP stream of their own, so their kernels
        overlap instead of queueing behind each other on the default stream."""
        if 0 < self.gpus_per < 1 and self.gpu_ids:
            import torch
            from ..train import graphed
            with self._lock:
                if self._shared_flag_before is None:
                    self._shared_flag_before = bool(graphed.SHARED_GPU_THREADS)
                graphed.SHARED_GPU_THREADS = True      # graph capture is not safe while sibling threads use the device;
            return torch.cuda.stream(torch.cuda.Stream())      # shutdown() puts the flag back
        return contextlib.nullcontext()

    def start_trial(self, trial: Trial) -> None:
        reply: "queue.Queue[str]" = queue.Queue()
        self._replies[trial.trial_id] = reply

        def send(msg: Dict) -> str:
            self._events.put(Event("result", trial, result=msg["result"], checkpoint=msg.get("checkpoint")))
            return reply.get()

        def body():
            sess = _session.Session(
                trial.trial_id, trial.config, send, checkpoint=trial.restore_from,
                logdir=trial.logdir, experiment_id=self.experiment_id,
                iteration_offset=trial.training_iteration,
                time_offset=float(trial.last_result.get("time_total_s", 0.0)))
            _session._set_session(sess)
            try:
                with self._stream_scope():
                    self.trainable(trial.config)
                self._events.put(Event("done", trial))
            except _session.StopTrial as st:
                self._events.put(Event("done", trial, result={"decision": st.decision}))
            except BaseException:  # noqa: BLE001 - the driver decides what a failure means
                self._events.put(Event("error", trial, error=traceback.format_exc()))
            finally:
                _session._set_session(None)

        th = threading.Thread(target=body, name=f"trial-{trial.trial_id}", daemon=True)
        self._threads[trial.trial_id] = th
        th.start()

    def get_events(self, timeout: Optional[float] = None) -> List[Event]:
        out: List[Event] = []
        try:
            out.append(self._events.get(timeout=timeout))
            while True:
                out.append(self._events.get_nowait())
        except queue.Empty:
            pass
        for ev in out:
            if ev.kind in ("done", "error"):
                th = self._threads.pop(ev.trial.trial_id, None)
                self._replies.pop(ev.trial.trial_id, None)
                if th is not None and th is not threading.current_thread():
                    th.join(timeout=5)
        return out

    def resolve(self, trial: Trial, decision: str) -> None:
        self._replies[trial.trial_id].put(decision)

    def shutdown(self) -> None:
        for tid, q in list(self._replies.items()):
            q.put(_session.STOP)
        for th in list(self._threads.values()):
            th.join(timeout=5)
        self._threads.clear()
        if self._shared_flag_before is not None:       # later runs in this process capture graphs again
            from ..train import graphed
            graphed.SHARED_GPU_THREADS = self._shared_flag_before
            self._shared_flag_before = None


# --------------------------------------------------------------------------- process executor
def _process_main(trainable, trial_id, config, conn, restore_from, logdir, experiment_id,
                  iteration_offset, time_offset):
    def send(msg: Dict) -> str:
        conn.send(msg)
        return conn.recv()

    sess = _session.Session(trial_id, config, send, checkpoint=restore_from, logdir=logdir,
                            experiment_id=experiment_id, iteration_offset=iteration_offset,
                            time_offset=time_offset)
    _session._set_session(sess, process_wide=True)
    try:
        trainable(config)
        conn.send({"type": "done"})
    except _session.StopTrial as st:
        conn.send({"type": "done", "decision": st.decision})
    except BaseException:  # noqa: BLE001
        conn.send({"type": "error", "error": traceback.format_exc()})
    finally:
        conn.close()


class ProcessExecutor(Executor):
    """One spawned worker process per running trial, GPU visibility narrowed per worker."""

    _env_lock = threading.Lock()

    def __init__(self, trainable: Callable, resources_per_trial: Optional[Dict] = None,
                 max_concurrent: Optional[int] = None, experiment_id: str = ""):
        self.trainable = trainable
        self.slots, self.gpu_ids, self.gpus_per = _resource_slots(resources_per_trial, max_concurrent)
        self.experiment_id = experiment_id
        self._ctx = mp.get_context("spawn")
        self._events: "queue.Queue[Event]" = queue.Queue()
        self._procs: Dict[str, Any] = {}
        self._conns: Dict[str, Any] = {}
        self._free_gpus = list(self.gpu_ids)
        self._held: Dict[str, List[int]] = {}

    def has_capacity(self) -> bool:
        return len(self._procs) < self.slots

    def num_running(self) -> int:
        return len(self._procs)

    def start_trial(self, trial: Trial) -> None:
        parent, child = self._ctx.Pipe()
        n_gpu = int(self.gpus_per) if self.gpus_per >= 1 else (1 if self.gpus_per > 0 else 0)
        mine: List[int] = []
        if self._free_gpus and n_gpu:
            if self.gpus_per >= 1:
                mine = [self._free_gpus.pop(0) for _ in range(min(n_gpu, len(self._free_gpus)))]
            else:   # fractional GPU: share round-robin without reserving
                mine = [self.gpu_ids[trial.index % len(self.gpu_ids)]]
        self._held[trial.trial_id] = mine if self.gpus_per >= 1 else []
        trial.placement["gpu"] = mine
        p = self._ctx.Process(
            target=_process_main,
            args=(self.trainable, trial.trial_id, trial.config, child, trial.restore_from, trial.logdir,
                  self.experiment_id, trial.training_iteration,
                  float(trial.last_result.get("time_total_s", 0.0))),
            daemon=True)
        with ProcessExecutor._env_lock:
            saved = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
            if mine:
                ids = ",".join(physical_device_ids(mine))      # through the mask this process itself runs under
                os.environ["HIP_VISIBLE_DEVICES"] = ids
                os.environ["CUDA_VISIBLE_DEVICES"] = ids
            try:
                p.start()
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        child.close()
        self._procs[trial.trial_id] = p
        self._conns[trial.trial_id] = parent

        def relay():
            try:
                while True:
                    msg = parent.recv()
                    if msg["type"] == "result":
                        self._events.put(Event("result", trial, result=msg["result"],
                                               checkpoint=msg.get("checkpoint")))
                    elif msg["type"] == "done":
                        self._events.put(Event("done", trial, result={"decision": msg.get("decision")}))
                        return
                    else:
                        self._events.put(Event("error", trial, error=msg.get("error")))
                        return
            except (EOFError, OSError):
                self._events.put(Event("error", trial, error="trial process died without a message "
                                                             f"(exit code {p.exitcode})"))

        threading.Thread(target=relay, daemon=True).start()

    def get_events(self, timeout: Optional[float] = None) -> List[Event]:
        out: List[Event] = []
        try:
            out.append(self._events.get(timeout=timeout))
            while True:
                out.append(self._events.get_nowait())
        except queue.Empty:
            pass
        for ev in out:
            if ev.kind in ("done", "error"):
                tid = ev.trial.trial_id
                p = self._procs.pop(tid, None)
                c = self._conns.pop(tid, None)
                self._free_gpus.extend(self._held.pop(tid, []))
                if p is not None:
                    p.join(timeout=10)
                    if p.is_alive():
                        p.terminate()
                if c is not None:
                    c.close()
        return out

    def resolve(self, trial: Trial, decision: str) -> None:
        try:
            self._conns[trial.trial_id].send(decision)
        except (KeyError, OSError, BrokenPipeError):
            pass

    def shutdown(self) -> None:
        for tid, p in list(self._procs.items()):
            if p.is_alive():
                p.terminate()
            p.join(timeout=5)
        self._procs.clear()
