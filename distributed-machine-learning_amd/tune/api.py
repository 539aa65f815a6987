"""``run`` / ``with_parameters`` -- the experiment entry points.

Call-compatible with how the reference drives its experiments
(/root/reference/ray-tune-hpo-regression.py:469-478,
/root/reference/ray-tune-hpo-regression-sample.py:159-167):

    analysis = run(with_parameters(train_fn, data_loader=tr, val_loader=va),
                   config=search_space, num_samples=10,
                   scheduler=ASHAScheduler(metric="validation_loss", mode="min"),
                   resources_per_trial={"cpu": 2, "gpu": 1}, local_dir="...", verbose=1)
    print(analysis.best_config)

Differences: no cluster runtime is needed; metric/mode are inherited from the scheduler or the
search algorithm when not passed here; GPU trials fall back to the CPU on a GPU-less machine
instead of pending forever; a ``PackedTrainable`` (see ``mitune.packed``) is executed as packs of
many trials per GPU instead of one process per trial.
"""
from __future__ import annotations

import functools
import inspect
import json
import os
import sys
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

from .analysis import ExperimentAnalysis
from .executor import Executor, ProcessExecutor, ThreadExecutor
from .logger import DEFAULT_LOGGERS, Logger
from .tbevents import TBXLogger
from .runner import TrialRunner, TuneError
from .trial import ERROR, PAUSED, PENDING, RUNNING, TERMINATED
from ..schedulers.base import FIFOScheduler, TrialScheduler
from ..search.searcher import BasicVariantGenerator, Searcher


def with_parameters(trainable: Callable, **kwargs) -> Callable:
    """Bind large read-only objects (datasets, loaders) to a trainable once.

    The reference wraps its training function this way so the data loaders are shipped to trial
    workers through Ray's object store rather than pickled per trial
    (/root/reference/ray-tune-hpo-regression.py:470).  With thread trials the objects are simply
    shared; with process trials torch tensors travel through shared memory; with packed trials
    the dataset becomes one device-resident copy per GPU used by every trial of the pack.
    """
    if not callable(trainable):
        raise TypeError("with_parameters() needs a callable trainable")
    inner = functools.partial(trainable, **kwargs)
    functools.update_wrapper(inner, trainable)
    inner._mitune_parameters = kwargs          # noqa: SLF001 - introspected by executors
    return inner


class ProgressReporter:
    """Compact status table printed every ``period_s`` seconds (``verbose >= 1``)."""

    def __init__(self, verbose: int = 1, period_s: float = 5.0, metric: Optional[str] = None,
                 mode: Optional[str] = None, stream=None, max_rows: int = 12):
        self.verbose, self.period_s, self.metric, self.mode = verbose, period_s, metric, mode
        self.stream = stream or sys.stdout
        self.max_rows = max_rows
        self._last = 0.0

    def maybe_report(self, runner: TrialRunner) -> None:
        if self.verbose >= 1 and time.time() - self._last >= self.period_s:
            self.report(runner)

    def report(self, runner: TrialRunner, final: bool = False) -> None:
        if self.verbose < 1:
            return
        self._last = time.time()
        counts: Dict[str, int] = {}
        for t in runner.trials:
            counts[t.status] = counts.get(t.status, 0) + 1
        el = time.time() - runner._t0
        done = counts.get(TERMINATED, 0) + counts.get(ERROR, 0)
        line = (f"== mitune {'final ' if final else ''}status: {el:7.1f}s | "
                + " ".join(f"{k}={v}" for k, v in sorted(counts.items()))
                + (f" | {done / el * 3600:.0f} trials/h" if el > 0 and done else ""))
        print(line, file=self.stream)
        if self.verbose >= 2 or final:
            metric = self.metric or runner.metric
            rows = [t for t in runner.trials if t.last_result]
            if metric:
                sign = -1 if (self.mode or runner.mode) == "max" else 1
                rows.sort(key=lambda t: sign * t.last_result.get(metric, float("inf"))
                          if isinstance(t.last_result.get(metric), (int, float)) else float("inf"))
            for t in rows[: self.max_rows]:
                mv = t.last_result.get(metric) if metric else None
                print(f"   {t.trial_id} {t.status:<10} it={t.training_iteration:<4}"
                      + (f" {metric}={mv:.6g}" if isinstance(mv, (int, float)) else ""), file=self.stream)
        if self.verbose >= 2:
            print("   " + runner.scheduler.debug_string(), file=self.stream)
        self.stream.flush()


def _trainable_name(trainable: Any) -> str:
    fn = trainable
    while isinstance(fn, functools.partial):
        fn = fn.func
    return getattr(fn, "__name__", None) or getattr(trainable, "name", None) or type(trainable).__name__


def run(run_or_experiment: Any,
        config: Optional[Dict] = None,
        num_samples: int = 1,
        scheduler: Optional[TrialScheduler] = None,
        search_alg: Optional[Searcher] = None,
        metric: Optional[str] = None,
        mode: Optional[str] = None,
        resources_per_trial: Optional[Dict] = None,
        local_dir: Optional[str] = None,
        name: Optional[str] = None,
        stop: Any = None,
        verbose: int = 1,
        max_failures: int = 0,
        raise_on_failed_trial: bool = True,
        fail_fast: bool = False,
        resume: Union[bool, str] = False,
        loggers: Optional[Sequence[Union[Logger, type]]] = None,
        keep_checkpoints_num: Optional[int] = None,
        checkpoint_at_end: bool = False,
        max_concurrent_trials: Optional[int] = None,
        time_budget_s: Optional[float] = None,
        executor: Union[None, str, Executor] = None,
        seed: Optional[int] = 0,
        progress_period_s: float = 5.0,
        **executor_kwargs) -> ExperimentAnalysis:
    """Run a hyper-parameter search and return its ``ExperimentAnalysis``.

    ``run_or_experiment`` is a function ``fn(config)`` (optionally wrapped by
    ``with_parameters``) that calls ``mitune.report(...)``, or a ``PackedTrainable`` describing a
    model family this framework can train as packs of trials on a GPU.

    ``executor``: ``"thread"`` (default for functions), ``"process"`` (one spawned worker per
    trial, GPUs pinned through ``HIP_VISIBLE_DEVICES``), ``"packed"`` (default for
    ``PackedTrainable``) or an ``Executor`` instance.
    """
    config = config or {}
    trainable = run_or_experiment
    tname = _trainable_name(trainable)

    # --- metric / mode: explicit > scheduler > searcher -------------------------------------
    for src in (scheduler, search_alg):
        if src is not None:
            metric = metric or getattr(src, "metric", None)
            mode = mode or getattr(src, "mode", None)
    if mode not in (None, "min", "max"):
        raise ValueError("mode must be 'min' or 'max'")
    scheduler = scheduler or FIFOScheduler()
    if not scheduler.set_search_properties(metric, mode):
        raise ValueError("scheduler was built with a different metric/mode than run()")

    if search_alg is None:
        search_alg = BasicVariantGenerator(num_samples=num_samples, seed=seed)
        max_trials = None
    else:
        max_trials = num_samples
    search_alg.set_search_properties(metric, mode, config)

    # --- executor ---------------------------------------------------------------------------
    from ..packed.trainable import PackedTrainable      # local import: packed pulls in torch
    if isinstance(executor, Executor):
        ex = executor
    elif isinstance(trainable, PackedTrainable):
        if executor not in (None, "packed"):
            raise ValueError("a PackedTrainable runs on the packed executor")
        from ..packed.executor import PackedExecutor
        ex = PackedExecutor(trainable, scheduler=scheduler, resources_per_trial=resources_per_trial,
                            max_concurrent=max_concurrent_trials, experiment_id=tname,
                            seed=seed, **executor_kwargs)
    elif executor in (None, "thread"):
        ex = ThreadExecutor(trainable, resources_per_trial, max_concurrent_trials, tname)
    elif executor == "process":
        ex = ProcessExecutor(trainable, resources_per_trial, max_concurrent_trials, tname)
    else:
        raise ValueError(f"unknown executor {executor!r}")
    if hasattr(ex, "adapt_to_space"):
        ex.adapt_to_space(config)          # an engine picked from the trainable alone may not hold every shape of this space
    # One process per GPU runs this same function.  Rank 0 is the primary: it names the experiment, owns the loggers
    # and experiment_state.json; every rank shares the directory (trial owners write their checkpoints into it).
    rank, world = int(getattr(ex, "rank", 0)), int(getattr(ex, "world", 1))
    comm = getattr(ex, "comm", None)

    # --- experiment directory ---------------------------------------------------------------
    experiment_dir = None
    if local_dir and rank == 0:
        local_dir = os.path.expanduser(local_dir)
        if resume and name is None:
            cands = sorted(d for d in os.listdir(local_dir) if d.startswith(tname + "_")) if os.path.isdir(local_dir) else []
            name = cands[-1] if cands else None
        name = name or f"{tname}_{time.strftime('%Y-%m-%d_%H-%M-%S')}"
    if world > 1:
        # rank 0's directory and name are THE directory and name: a time-stamped default differs between ranks started a
        # second apart, and ranks with different trees would checkpoint and resume different searches
        local_dir, name = comm.broadcast_object((local_dir, name) if rank == 0 else None, src=0)
    if local_dir:
        experiment_dir = os.path.join(local_dir, name)
        os.makedirs(experiment_dir, exist_ok=True)
    experiment_id = name or tname
    ex.experiment_id = experiment_id

    def make_loggers() -> List[Logger]:
        return [lg() if isinstance(lg, type) else lg
                for lg in (loggers if loggers is not None else (DEFAULT_LOGGERS + (TBXLogger,) if experiment_dir else ()))]

    # result.json / progress.csv / tfevents have one writer: rank 0 (a rank that becomes rank 0 builds its own then)
    logger_objs: List[Logger] = make_loggers() if rank == 0 else []

    runner = TrialRunner(
        ex, search_alg, scheduler, logger_objs, trainable_name=tname, experiment_dir=experiment_dir,
        metric=metric, mode=mode, stop=stop, max_failures=max_failures,
        keep_checkpoints_num=keep_checkpoints_num, checkpoint_at_end=checkpoint_at_end,
        max_trials=max_trials, progress=ProgressReporter(verbose, progress_period_s, metric, mode),
        time_budget_s=time_budget_s, fail_fast=fail_fast, is_primary=(rank == 0), logger_factory=make_loggers)
    if hasattr(ex, "bind_runner"):
        ex.bind_runner(runner)

    if resume and experiment_dir and os.path.exists(os.path.join(experiment_dir, "experiment_state.json")):
        with open(os.path.join(experiment_dir, "experiment_state.json")) as f:
            runner.restore_state(json.load(f), resume_errored=(resume == "ERRORED_ONLY" or resume == "ALL"))

    runner.run()

    analysis = ExperimentAnalysis(runner.trials, metric, mode, experiment_dir,
                                  stats={**runner.stats, "elapsed_s": time.time() - runner._t0,
                                         **getattr(ex, "stats", {})})
    failed = [t for t in runner.trials if t.status == ERROR]
    if failed and raise_on_failed_trial:
        msg = "\n".join(f"--- trial {t.trial_id} ---\n{t.error_msg}" for t in failed[:3])
        err = TuneError(f"{len(failed)} trial(s) failed:\n{msg}")
        err.analysis = analysis
        raise err
    return analysis
