"""TrialRunner: the driver loop between searcher, scheduler, executor and loggers.

This is the part of ``tune.run`` the reference gets from Ray
(/root/reference/ray-tune-hpo-regression.py:469-478): create ``num_samples`` trials from the
search space, keep as many running as resources allow, route every reported result through the
scheduler (CONTINUE / STOP / PAUSE / RESTART) and the searcher, retry failed trials up to
``max_failures`` times from their newest checkpoint, and persist enough state to resume.
"""
from __future__ import annotations

import collections
import os
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

from . import checkpoint as ckpt
from .executor import Event, Executor
from .logger import Logger, dump_json
from .session import CONTINUE, PAUSE, RESTART, STOP
from .trial import ERROR, PAUSED, PENDING, RUNNING, TERMINATED, Trial
from ..schedulers.base import FIFOScheduler, TrialScheduler
from ..search.searcher import Searcher


class TuneError(RuntimeError):
    pass


StopCriterion = Union[None, Dict[str, float], Callable[[str, Dict], bool]]


class TrialRunner:
    def __init__(self, executor: Executor, searcher: Searcher, scheduler: Optional[TrialScheduler] = None,
                 loggers: Sequence[Logger] = (), trainable_name: str = "trainable",
                 experiment_dir: Optional[str] = None, metric: Optional[str] = None, mode: Optional[str] = None,
                 stop: StopCriterion = None, max_failures: int = 0, keep_checkpoints_num: Optional[int] = None,
                 checkpoint_at_end: bool = False, max_trials: Optional[int] = None,
                 progress: Optional[Any] = None, state_save_period_s: float = 10.0,
                 time_budget_s: Optional[float] = None, fail_fast: bool = False, is_primary: bool = True,
                 logger_factory: Optional[Callable[[], List[Logger]]] = None):
        self.executor, self.searcher = executor, searcher
        self.scheduler = scheduler or FIFOScheduler()
        self.loggers = list(loggers)
        self.trainable_name, self.experiment_dir = trainable_name, experiment_dir
        self.metric, self.mode = metric, mode
        self.stop, self.max_failures = stop, max_failures
        self.keep_checkpoints_num, self.checkpoint_at_end = keep_checkpoints_num, checkpoint_at_end
        self.max_trials = max_trials
        self.progress = progress
        self.state_save_period_s = state_save_period_s
        self.time_budget_s, self.fail_fast = time_budget_s, fail_fast
        # one process per GPU runs an identical runner; only the primary (rank 0) writes experiment_state.json
        self.is_primary = bool(is_primary)
        self.logger_factory = logger_factory     # builds the loggers on a rank that BECOMES primary (rank 0 was lost)
        self.trials: List[Trial] = []
        self._by_id: Dict[str, Trial] = {}
        self._pending_q: "collections.deque[Trial]" = collections.deque()    # PENDING trials in arrival order
        self.num_paused = 0
        self._unfinished = 0
        self._searcher_done = False
        self._t0 = time.time()
        self._last_save = 0.0
        self._pending_decision: Dict[str, str] = {}
        self.executor_moves_weights = bool(getattr(executor, "moves_weights", False))
        self.stats: Dict[str, Any] = {"num_results": 0, "num_stopped_early": 0, "num_errors": 0,
                                      "num_restarts": 0}

    # ------------------------------------------------------------------ trial bookkeeping
    def add_trial(self, trial: Trial) -> None:
        self.trials.append(trial)
        self._by_id[trial.trial_id] = trial
        if trial.status == PENDING:
            self._pending_q.append(trial)
        elif trial.status == PAUSED:
            self.num_paused += 1
        if not trial.is_finished:
            self._unfinished += 1
        self.scheduler.on_trial_add(self, trial)

    def _set_status(self, trial: Trial, status: str) -> None:
        """Status change with the O(1) bookkeeping that keeps scheduling cost linear in the number of trials."""
        was_done, was_paused = trial.is_finished, trial.status == PAUSED
        trial.set_status(status)
        if status == PENDING:
            self._pending_q.append(trial)
        self.num_paused += int(status == PAUSED) - int(was_paused)
        self._unfinished += int(was_done) - int(trial.is_finished)

    def next_pending(self) -> Optional[Trial]:
        q = self._pending_q
        while q and q[0].status != PENDING:
            q.popleft()
        return q[0] if q else None

    def lookahead(self, n: int) -> List[Trial]:
        """Up to ``n`` trials that will start next, in start order, without starting them: the pending queue, topped up
        from the searcher when its suggestions do not depend on results (``lookahead_safe``).  Executors use it to
        prepare trials while the device is busy."""
        out = [t for t in self._pending_q if t.status == PENDING][:n]
        if getattr(self.searcher, "lookahead_safe", False):
            while len(out) < n:
                t = self._next_from_searcher()
                if t is None:
                    break
                out.append(t)
        return out

    def get_trial(self, trial_id: str) -> Optional[Trial]:
        return self._by_id.get(trial_id)

    def _next_from_searcher(self) -> Optional[Trial]:
        if self._searcher_done or (self.max_trials is not None and len(self.trials) >= self.max_trials):
            self._searcher_done = True
            return None
        index = len(self.trials)
        trial_id = f"{index:05d}"
        cfg = self.searcher.suggest(trial_id)
        if cfg is Searcher.FINISHED:
            self._searcher_done = True
            return None
        if cfg is None:
            return None
        t = Trial(cfg, trial_id=trial_id, index=index, trainable_name=self.trainable_name,
                  experiment_dir=self.experiment_dir)
        self.add_trial(t)
        return t

    # ------------------------------------------------------------------ main loop
    def is_finished(self) -> bool:
        if not self._searcher_done:
            return False
        return self._unfinished == 0

    def _start_trials(self) -> None:
        while self.executor.has_capacity():
            t = self.scheduler.choose_trial_to_run(self)
            if t is None:
                t = self._next_from_searcher()
                if t is None:
                    break
                continue    # let the scheduler pick (normally it picks the trial just added)
            self._launch(t)

    def _launch(self, trial: Trial) -> None:
        fresh = trial.start_time is None
        self._set_status(trial, RUNNING)
        if fresh:
            for lg in self.loggers:
                lg.on_trial_start(trial)
        self.executor.start_trial(trial)

    def step(self) -> None:
        self._start_trials()
        if self.executor.num_running() == 0:
            if self.is_finished():
                return
            if self._searcher_done:
                # nothing runs and the scheduler resumes none of the parked trials: close them
                for t in self.trials:
                    if t.status in (PENDING, PAUSED):
                        self._finish(t, TERMINATED)
                return
            self._idle_spins = getattr(self, "_idle_spins", 0) + 1
            if self._idle_spins > 200:
                raise TuneError("searcher suggests nothing while no trial is running")
            time.sleep(0.01)
            return
        self._idle_spins = 0
        events = self.executor.get_events(timeout=0.5)
        results = [e for e in events if e.kind == "result"]
        if results and getattr(self.executor, "lockstep", False):
            self._on_batch(results)
        else:
            for e in results:
                self._on_result(e)
        for e in events:
            if e.kind == "done":
                self._on_done(e)
            elif e.kind == "error":
                self._on_error(e)
            elif e.kind == "lost":
                self._on_lost(e)
        now = time.time()
        if self.experiment_dir and now - self._last_save > self.state_save_period_s:
            self.save_state()
        if self.progress is not None:
            self.progress.maybe_report(self)
        # A multi-rank executor supplies a clock all ranks agree on (sum of gathered round times); a rank's own clock
        # would end the search on one rank a round earlier than on another.
        elapsed = getattr(self.executor, "replicated_elapsed_s", None)
        if elapsed is None:
            elapsed = now - self._t0
        if self.time_budget_s is not None and elapsed > self.time_budget_s:
            self._searcher_done = True
            for t in self.trials:
                if t.status in (PENDING, PAUSED):
                    self._finish(t, TERMINATED)

    def run(self) -> None:
        try:
            while not self.is_finished():
                self.step()
        finally:
            self.executor.shutdown()
            for lg in self.loggers:
                lg.close()
            if self.experiment_dir:
                self.save_state()
            if self.progress is not None:
                self.progress.report(self, final=True)

    # ------------------------------------------------------------------ event handlers
    def decides_at(self, iteration: int) -> bool:
        """False when a result at this training iteration can only be answered CONTINUE: no stop criterion, no time
        budget, and a scheduler that decides nothing there."""
        if self.stop is not None or self.time_budget_s is not None:
            return True
        return bool(self.scheduler.decides_at(iteration))

    def _should_stop(self, trial: Trial, result: Dict) -> bool:
        if self.stop is None:
            return False
        if callable(self.stop):
            return bool(self.stop(trial.trial_id, result))
        for k, thr in self.stop.items():
            if k in result and result[k] >= thr:
                return True
        return False

    def _record(self, ev: Event) -> None:
        trial, result = ev.trial, ev.result
        result.setdefault("config", trial.config)
        trial.record_result(result)
        self.stats["num_results"] += 1
        if ev.checkpoint is not None:
            trial.checkpoint = ev.checkpoint
            trial.checkpoint_iter = result.get("training_iteration", trial.checkpoint_iter + 1)
            if trial.logdir:
                ckpt.save_checkpoint(trial.logdir, trial.checkpoint_iter, ev.checkpoint, self.keep_checkpoints_num)
        for lg in self.loggers:
            lg.on_result(trial, result)
        self.searcher.on_trial_result(trial.trial_id, result)

    def _apply(self, trial: Trial, decision: str, result: Dict) -> None:
        if self._should_stop(trial, result):
            decision = STOP
        elif decision == STOP:
            max_t = getattr(self.scheduler, "max_t", None)
            if max_t is None or result.get("training_iteration", 0) < max_t:     # a trial that ran its full budget
                self.stats["num_stopped_early"] += 1                              # was not stopped early
        self._pending_decision[trial.trial_id] = decision
        self.executor.resolve(trial, decision)

    def _on_result(self, ev: Event) -> None:
        self._record(ev)
        decision = self.scheduler.on_trial_result(self, ev.trial, ev.result)
        self._apply(ev.trial, decision, ev.result)

    def _on_batch(self, events: List[Event]) -> None:
        for ev in events:
            self._record(ev)
        decisions = self.scheduler.on_batch_result(self, [e.trial for e in events], [e.result for e in events])
        for ev, d in zip(events, decisions):
            self._apply(ev.trial, d, ev.result)

    def _finish(self, trial: Trial, status: str) -> None:
        if trial.last_result:
            trial.last_result["done"] = True
        self._set_status(trial, status)
        if status == TERMINATED:
            self.scheduler.on_trial_complete(self, trial, trial.last_result)
            self.searcher.on_trial_complete(trial.trial_id, trial.last_result, error=False)
        else:
            self.scheduler.on_trial_error(self, trial)
            self.searcher.on_trial_complete(trial.trial_id, None, error=True)
        for lg in self.loggers:
            lg.on_trial_end(trial)

    def _on_done(self, ev: Event) -> None:
        trial = ev.trial
        decision = (ev.result or {}).get("decision") or self._pending_decision.get(trial.trial_id)
        self._pending_decision.pop(trial.trial_id, None)
        if decision == RESTART and getattr(trial, "pbt_request", None):
            req = trial.pbt_request
            trial.pbt_request = None
            donor = self._by_id.get(req["donor"])
            trial.config = req["config"]
            trial.restore_from = donor.checkpoint if donor is not None else trial.checkpoint
            self._set_status(trial, PENDING)
            self.stats["num_restarts"] += 1
            if trial.logdir:
                dump_json(trial.config, os.path.join(trial.logdir, "params.json"))
            return
        if decision == PAUSE:
            trial.restore_from = trial.checkpoint
            trial.restore_pending = True
            self._set_status(trial, PAUSED)
            return
        self._finish(trial, TERMINATED)

    def _on_error(self, ev: Event) -> None:
        trial = ev.trial
        trial.num_failures += 1
        trial.error_msg = ev.error
        self.stats["num_errors"] += 1
        self._pending_decision.pop(trial.trial_id, None)
        if self.fail_fast:
            self._finish(trial, ERROR)
            raise TuneError(f"trial {trial.trial_id} failed (fail_fast):\n{ev.error}")
        if self.max_failures < 0 or trial.num_failures <= self.max_failures:
            trial.restore_from = trial.checkpoint
            trial.restore_pending = True          # (a multi-rank executor fetches the state from the rank that holds it)
            self._set_status(trial, PENDING)      # retried from its newest checkpoint
            return
        self._finish(trial, ERROR)

    def _on_lost(self, ev: Event) -> None:
        """The rank that trained this trial is gone (not the trial's fault: no failure is counted).  It goes back to the
        queue and restarts on a surviving rank from its newest checkpoint on disk -- the experiment directory is shared by
        the ranks of a node -- or, without one, from its initial weights: training is deterministic in (seed, trial), so
        the restarted trial re-reports the iterations it had already reported with the same numbers, and schedulers that
        record one value per rung ignore the repeats."""
        trial = ev.trial
        self._pending_decision.pop(trial.trial_id, None)
        self.stats["num_lost"] = self.stats.get("num_lost", 0) + 1
        cp = ckpt.latest_checkpoint(trial.logdir) if trial.logdir else None
        it = 0
        if cp:
            try:
                it = int(ckpt.load_checkpoint(cp).get("training_iteration", 0))
            except Exception:       # noqa: BLE001 - a checkpoint cut off by the crash counts as none
                it = 0
        comm = getattr(self.executor, "comm", None)
        if comm is not None and getattr(comm, "world_size", 1) > 1:
            # ranks on different nodes may not see the same files: the (new) rank 0's reading of the disk stands for all,
            # so that every rank truncates the trial's history at the same iteration
            it = int(comm.broadcast_object(it, src=0))
        trial.checkpoint, trial.checkpoint_iter, trial.checkpoint_rank = None, it, None
        trial.restore_from, trial.restore_pending = None, True
        trial.results = [r for r in trial.results if r.get("training_iteration", 0) <= it]
        trial.last_result = trial.results[-1] if trial.results else {}
        self._set_status(trial, PENDING)

    def on_ranks_lost(self, is_primary: bool) -> None:
        """The executor re-formed its group without some ranks.  If the primary was among them, the new rank 0 takes over
        ``experiment_state.json`` and the loggers (which append to the files the old primary wrote)."""
        if is_primary and not self.is_primary:
            self.is_primary = True
            if self.logger_factory is not None and not self.loggers:
                self.loggers = list(self.logger_factory())

    # ------------------------------------------------------------------ persistence
    def save_state(self) -> None:
        if not self.experiment_dir or not self.is_primary:
            return
        os.makedirs(self.experiment_dir, exist_ok=True)
        state = {
            "trials": [t.to_json() for t in self.trials],
            "metric": self.metric, "mode": self.mode, "trainable_name": self.trainable_name,
            "scheduler": self.scheduler.state_dict(), "searcher": self.searcher.state_dict(),
            "searcher_done": self._searcher_done, "stats": self.stats,
            "timestamp": time.time(), "elapsed_s": time.time() - self._t0,
        }
        dump_json(state, os.path.join(self.experiment_dir, "experiment_state.json"))
        self._last_save = time.time()

    def restore_state(self, state: Dict, resume_errored: bool = False) -> None:
        """Rebuild the trial table from ``experiment_state.json``.

        Finished trials stay finished; trials that were running or pending when the previous
        driver died go back to PENDING and restart from their newest on-disk checkpoint.
        """
        self.searcher.load_state_dict(state.get("searcher", {}))
        self.scheduler.load_state_dict(state.get("scheduler", {}))
        self._searcher_done = bool(state.get("searcher_done", False))
        for tj in state["trials"]:
            t = Trial.from_json(tj)
            if t.status in (RUNNING, PENDING, PAUSED) or (resume_errored and t.status == ERROR):
                t.status = PENDING
                t.start_time = tj.get("start_time")
                cp = ckpt.latest_checkpoint(t.logdir) if t.logdir else None
                if cp:
                    t.restore_from = ckpt.load_checkpoint(cp)
                    t.checkpoint = t.restore_from
                    # results after the checkpoint are replayed by the trainable; drop them
                    it = int(t.restore_from.get("training_iteration", 0))
                    t.results = [r for r in t.results if r.get("training_iteration", 0) <= it]
                    t.last_result = t.results[-1] if t.results else {}
                else:
                    t.results, t.last_result = [], {}
            self.trials.append(t)
            self._by_id[t.trial_id] = t
            if not t.is_finished:
                self._unfinished += 1
                self._pending_q.append(t)
                self.scheduler.on_trial_add(self, t)
