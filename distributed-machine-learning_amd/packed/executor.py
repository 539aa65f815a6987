"""PackedExecutor: trials as slots of a per-GPU pack, advanced one epoch per scheduling round.

Replaces the reference's "one Ray actor + one GPU per trial" placement
(/root/reference/ray-tune-hpo-regression.py:475).  Every rank (one process per GPU) runs this
executor under an identical ``TrialRunner``: searcher and scheduler are replicated and seeded
alike, so all ranks create the same trials in the same order.  A trial is *owned* by one rank
(the one with the most free slots when it starts) and only its owner trains it; once early stopping
has thinned the GPUs unevenly and no pending trial is left to fill the holes, trials move from the
fullest GPU to the emptiest (their slab is broadcast over the job's communicator), because a round
lasts as long as its fullest GPU.  After each epoch
the ranks all-gather one ``[capacity, 6]`` record table (RCCL over xGMI on a GPU node); every rank
then feeds the same trial-id-sorted batch of results to its scheduler and reaches the same
CONTINUE / STOP / RESTART decisions without a coordinator.  Stopped trials free their slot at once
and the next pending trials take it, so early stopping turns into throughput.
"""
from __future__ import annotations

import math
import os
import socket
import time
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import PackEngine, TorchPackEngine
from .trainable import PackedTrainable
from ..options import OPTIONS
from ..parallel.comm import Comm, default_comm
from ..tune import checkpoint as ckpt
from ..tune.executor import Event, Executor
from ..tune.logger import dump_json
from ..tune.session import CONTINUE, PAUSE, RESTART, STOP
from ..tune.trial import Trial

class RankLostError(RuntimeError):
    """The per-epoch exchange between ranks failed (a peer process is gone or the fabric timed out) and the group could
    not be re-formed from the survivors, or the trainable asks for ``on_rank_loss="raise"``.  The runner saves the
    experiment state on the way out; restart every rank with ``resume=True``."""


_REC = 6   # trial_index, epoch, train_loss, validation_loss, validation_mse, validation_mape


def make_engine(trainable: PackedTrainable, device: torch.device, capacity: Optional[int], seed: int) -> PackEngine:
    kind = trainable.engine
    if kind == "auto":
        # the grouped-GEMM engine packs MLP trials; other families run slot by slot on the same device, each
        # slot's training step captured in a hipGraph (fused attention / LayerNorm / optimiser kernels inside)
        if device.type == "cuda":
            kind = "hip" if trainable.model == "mlp" else "graph"
            if trainable.model == "simple_transformer" and trainable.dtype == "f32" \
                    and OPTIONS.seq_engine:
                # small encoder trials share launches (grouped GEMMs, grouped attention / LayerNorm); shapes the grouped
                # kernels do not cover stay on the one-graph-per-slot engine
                from . import engine_seq
                if engine_seq.supports(trainable) is None:
                    kind = "seq"
            if trainable.model == "transformer" and OPTIONS.xf_engine:
                # the configurable Transformer family: one program of grouped launches for the whole pack
                from . import engine_xf
                if engine_xf.supports(trainable) is None:
                    kind = "xf"
        else:
            kind = "torch"
    if kind == "xf":
        from .engine_xf import XfPackEngine
        return XfPackEngine(trainable, device, capacity, seed)
    if kind == "seq":
        from .engine_seq import SeqPackEngine
        return SeqPackEngine(trainable, device, capacity, seed)
    if kind == "graph":
        from .engine_graph import GraphPackEngine
        return GraphPackEngine(trainable, str(device), capacity, seed)
    if kind == "torch":
        return TorchPackEngine(trainable, str(device), capacity, seed)
    if kind == "hip":
        from .engine_hip import HipPackEngine
        return HipPackEngine(trainable, device, capacity, seed)
    raise ValueError(f"unknown engine {kind!r}")


class _Ready:
    """A finished 'collective' (one rank): ``wait()`` returns the value."""

    def __init__(self, value: torch.Tensor):
        self.value = value

    def wait(self) -> torch.Tensor:
        return self.value


class PackedExecutor(Executor):
    lockstep = True          # results of one epoch form one scheduler batch
    moves_weights = True     # PBT exploits are slab copies / send-recv, not checkpoint restarts

    def __init__(self, trainable: PackedTrainable, scheduler=None, resources_per_trial: Optional[Dict] = None,
                 max_concurrent: Optional[int] = None, experiment_id: str = "", seed: Optional[int] = 0,
                 comm: Optional[Comm] = None, engine: Optional[PackEngine] = None,
                 device: Optional[torch.device] = None, capacity: Optional[int] = None):
        self.t = trainable
        self.comm = comm or default_comm()
        self.rank, self.world = self.comm.rank, self.comm.world_size
        self.device = device or self.comm.device
        self.seed = int(seed or 0)
        self.engine = engine or make_engine(trainable, self.device, capacity or trainable.capacity, self.seed)
        self.capacity = self.engine.capacity
        self.max_concurrent = max_concurrent
        self.experiment_id = experiment_id
        self._free: List[List[int]] = [list(range(self.capacity)) for _ in range(self.world)]
        self._place: Dict[str, Tuple[int, int]] = {}          # trial_id -> (rank, slot)
        # what every live trial weighs (engine.trial_cost of its configuration: replicated, like the placement) and the
        # sum per rank: trials are dealt and moved so that the most loaded rank -- which sets the round time -- is as light
        # as possible.  An engine that weighs every trial 1.0 gets the plain count balance.
        self._cost: Dict[str, float] = {}
        self._load: List[float] = [0.0] * self.world
        self._live: Dict[str, Trial] = {}
        self._queued: List[Event] = []
        self._awaiting: Dict[str, Optional[str]] = {}          # trial_id -> decision (None = not yet)
        self._epoch_time = 0.0
        # Seconds of search as every rank counts them alike: the sum over rounds of the slowest rank's round time, taken
        # from the gathered table.  Anything that decides something (time budgets, stop criteria on time fields,
        # schedulers with a time attribute) reads this, never a rank's own clock: ranks that disagree on a decision
        # issue different collectives and hang each other.
        self.replicated_elapsed_s = 0.0
        self._slot_time: Optional[float] = None   # seconds per live trial of the costliest rank, last round (replicated)
        self._retired_last = 0             # slots freed by the previous round's decisions
        self._retired_now = 0
        self._n_undecided = 0
        # a lost rank is repaired in place only where a failed collective raises in the caller (parallel/comm.py)
        self._repairable = self.t.on_rank_loss == "repair" and bool(getattr(self.comm, "delivers_errors", True))
        self._hostname = socket.gethostname()
        self._diverged: set = set()
        self._lost_pending: List[Trial] = []      # trials of ranks lost while moving trials: reported with the next round
        self._t_total: Dict[str, float] = {}
        # Asynchronous rounds (trainable.exchange == "async"): the all-gather of a round's table is posted and the next
        # epoch is trained before anyone waits for it, so a rank waits for a slower one only once that one is a whole epoch
        # behind; results -- and with them every decision -- arrive one round late, on every rank alike.
        # "rungs" does the same only on rounds whose results cannot lead to a decision (the scheduler's ``decides_at``:
        # between the rungs of ASHA, between the perturbations of PBT) and reads everything that is in flight before a
        # round that can: the same results, decisions and work as lockstep, minus the waits in between.
        self.exchange = getattr(self.t, "exchange", "lockstep")
        self.async_rounds = self.exchange in ("async", "rungs")
        self._inflight: List[Tuple[Any, torch.Tensor, Tuple[float, float, float, int]]] = []   # posted, not yet read
        self._trained: Dict[str, int] = {}          # epochs a resident trial has trained (>= its reported iterations)
        self._draining: Dict[str, Trial] = {}       # trained to max_epochs, slot already freed, last result in flight
        self._must_read = False                     # "rungs": everything in flight is being read before the next epoch
        self.runner = None
        self.stats: Dict[str, Any] = {"epochs": 0, "slot_epochs": 0, "train_s": 0.0, "eval_s": 0.0,
                                      "comm_s": 0.0, "engine": self.engine.name, "world_size": self.world,
                                      "capacity_per_gpu": self.capacity, "exploits": 0, "diverged": 0,
                                      "migrations": 0, "migrate_s": 0.0}

        if self.world > 1 and self.t.rebalance_gap > 0 and hasattr(self.engine, "raw_numel"):
            # the first use of a collective kind sets up its connections: paid here, not inside a scheduling round
            self.comm.broadcast(torch.zeros(self.engine.raw_numel, dtype=torch.float32, device=self.device), src=0)

    def bind_runner(self, runner) -> None:
        self.runner = runner

    def adapt_to_space(self, config: Optional[Dict]) -> None:
        """Called by ``run()`` with the search space before the first trial starts.  The grouped-launch engines (``seq``
        for the sample Transformer, ``xf`` for the configurable one) are chosen from the trainable alone, but whether they
        can hold a trial depends on the trial's shape: every combination of the shape keys the space can produce is
        checked here, the engine's limits are set from them, and a space with a combination outside the grouped kernels'
        envelope (or with a non-enumerable domain on one of those keys) gets the one-graph-per-slot engine instead, which
        takes any shape."""
        if self.engine.name not in ("seq", "xf") or self.t.engine != "auto" or self._place or not config:
            return
        import itertools
        from ..tune.space import Categorical, Domain
        if self.engine.name == "seq":
            from . import engine_seq as mod
            keys = ("d_model", "num_heads", "num_layers", "batch_size")
            limits = (("d_model", "max_d_model"), ("num_heads", "max_heads"), ("num_layers", "max_layers"),
                      ("batch_size", "max_batch"))
            need = {"d_model", "num_heads"}
            base = {"num_layers": 1}
        else:
            from . import engine_xf as mod
            keys = ("d_model", "num_heads", "num_encoder_layers", "batch_size", "key_dim_scaling", "attention_type",
                    "attn_kernel_size")
            limits = (("d_model", "max_d_model"), ("num_encoder_layers", "max_layers"), ("batch_size", "max_batch"))
            need = {"d_model"}
            base = {}
        options: List[List[Any]] = []
        for k in keys:
            v = config.get(k, self.t.model_kwargs.get(k))
            if isinstance(v, Categorical):
                options.append(list(v.categories))
            elif isinstance(v, Domain) or callable(v) or isinstance(v, dict):
                options = []                     # a domain that cannot be listed: no guarantee possible
                break
            else:
                options.append([v])
        why = None if options else "the search space draws an engine shape from a domain that cannot be enumerated"
        combos = [dict(zip(keys, c)) for c in itertools.product(*options)] if options else []
        for c in combos:
            c = {k: v for k, v in c.items() if v is not None}
            why = mod.supports(self.t, {**base, **c}) if need <= set(c) else None
            if why:
                break
        if why is None and combos:
            kw = dict(self.t.model_kwargs)
            have = lambda k: [c[k] for c in combos if c.get(k) is not None]      # noqa: E731
            for k, limit in limits:
                if have(k):
                    kw[limit] = max(int(kw.get(limit, 0) or 0), max(int(x) for x in have(k)))
            if self.engine.name == "xf":
                ff = config.get("dim_feedforward", self.t.model_kwargs.get("dim_feedforward"))
                if isinstance(ff, Categorical):
                    kw["max_dim_feedforward"] = max(int(x) for x in ff.categories)
                elif isinstance(ff, (int, float)):
                    kw["max_dim_feedforward"] = int(ff)
                else:       # a multiple of d_model drawn per trial (the reference's space: 2, 3 or 4 times): the kernels' limit
                    kw["max_dim_feedforward"] = min(2048, 4 * int(kw.get("max_d_model", 512)))
            if kw != self.t.model_kwargs:
                self.t.model_kwargs = kw
                self.engine.close()
                self.engine = make_engine(self.t, self.device, self.capacity, self.seed)
            return
        if why is not None:
            self.engine.close()
            from .engine_graph import GraphPackEngine
            self.engine = GraphPackEngine(self.t, str(self.device), self.capacity, self.seed)
            self.capacity = self.engine.capacity
            self._free = [list(range(self.capacity)) for _ in range(self.world)]
            self.stats["engine"] = self.engine.name
            self.stats["engine_fallback"] = why

    # ------------------------------------------------------------------ capacity / placement
    def has_capacity(self) -> bool:
        if self.max_concurrent is not None and len(self._live) >= self.max_concurrent:
            return False
        return any(self._free[r] for r in range(self.world))

    def num_running(self) -> int:
        return len(self._live) + len(self._queued)

    def start_trial(self, trial: Trial) -> None:
        cost = float(self.engine.trial_cost(trial.config))
        # the least loaded rank that has a free slot (equal weights: the rank with the most free slots; ties: lowest rank)
        owner = min((r for r in range(self.world) if self._free[r]),
                    key=lambda r: (round(self._load[r], 12), -len(self._free[r]), r))
        slot = self._free[owner].pop(0)
        self._cost[trial.trial_id] = cost
        self._load[owner] += cost
        self._place[trial.trial_id] = (owner, slot)
        self._live[trial.trial_id] = trial
        self._trained[trial.trial_id] = int(trial.training_iteration)
        self._t_total.setdefault(trial.trial_id, float(trial.last_result.get("time_total_s", 0.0)))
        trial.placement = {"rank": owner, "slot": slot}
        state = trial.restore_from
        if getattr(trial, "restore_pending", False):
            # A retried or resumed trial continues from its newest checkpoint, which only the rank that owned it then
            # holds in memory.  Which rank that was is replicated knowledge (``checkpoint_rank``); if the trial now goes
            # to another rank, the state travels in one broadcast that every rank takes part in.
            trial.restore_pending = False
            src = getattr(trial, "checkpoint_rank", None)
            if self.world > 1 and src is not None and src != owner:
                state = self.comm.broadcast_object(state if self.rank == src else None, src=src)
        if owner == self.rank:
            if state is None and trial.checkpoint_iter > 0 and trial.logdir:      # e.g. a rank that never held it
                cp = ckpt.latest_checkpoint(trial.logdir)
                state = ckpt.load_checkpoint(cp) if cp else None
            self.engine.load(slot, trial.index, trial.config, state=state)
        trial.restore_from = None

    # ------------------------------------------------------------------ one scheduling round
    def get_events(self, timeout: Optional[float] = None) -> List[Event]:
        if self._queued:
            out, self._queued = self._queued, []
            return out
        if not self._live:
            self._discard_inflight()
            return []
        if self._n_undecided > 0:
            return []      # the runner still owes decisions for the previous round
        if self.async_rounds:
            return self._async_round()
        table, times = self._train_round()
        lost_trials, self._lost_pending = self._lost_pending, []
        gathered, lost = self._gather(table)
        return self._report(gathered, times, lost_trials + lost)

    def _train_round(self) -> Tuple[torch.Tensor, Tuple[float, float, float, int]]:
        """Level the ranks, train every resident trial for one epoch, validate, and build this rank's record table (one
        row per slot + one row of timings).  Returns it with (t0, end of training, end of validation, local trials)."""
        if self.world > 1 and self.t.rebalance_gap > 0:
            self._rebalance()      # the runner has filled what it could with pending trials; even out what is left
        t0 = time.time()
        device_table = getattr(self.engine, "device_table", None)
        if device_table is not None:
            # The engine writes the record table on the device (metrics, the round's duration from the GPU clock) and the
            # ranks all-gather that very buffer: between the last validation kernel and the collective nothing goes
            # through the host; the one synchronisation of the round is the read-back of the gathered table.
            self.engine.begin_round()
            self.engine.train_epoch()
            table = device_table()
            if self.async_rounds:
                table = table.clone()        # the next epoch rewrites the engine's buffer while this one is in flight
            if self.world == 1 and self.runner is not None and hasattr(self.engine, "prepare"):
                self._prefetch(t0)      # the epoch and its validation are enqueued: build the next trials meanwhile
            if self.comm.device.type != "cuda":       # a CPU-side collective (gloo, in-process ranks): one copy down
                table = table.cpu()
            t1 = t2 = time.time()
            n_local = len(self.engine.occupied())
        else:
            begin = getattr(self.engine, "train_epoch_begin", None)
            if begin is not None and self.world == 1 and self.runner is not None and hasattr(self.engine, "prepare"):
                begin()                                   # the GPU trains ...
                self._prefetch(t0)                        # ... while the host builds and captures the trials that come next
                self.engine.train_epoch_end()
            else:
                self.engine.train_epoch()
            t1 = time.time()
            local = self.engine.evaluate()
            t2 = time.time()
            n_local = len(local)
            mine = {slot: tid for tid, (r, slot) in self._place.items() if r == self.rank}
            rows_local = [[-1.0] * _REC for _ in range(self.capacity)]
            for slot, m in local.items():
                tr = self._live[mine[slot]]
                rows_local[slot] = [float(tr.index), float(self._trained[tr.trial_id] + 1), m["train_loss"],
                                    m["validation_loss"], m["validation_mse"], m["validation_mape"]]
            # one more row: this rank's time for the round and how many trials it spent it on; gathered with the metrics,
            # so that every rank knows every rank's cost per trial and all of them weigh a move the same way
            rows_local.append([-2.0, t2 - t0, float(len(local)), t1 - t0, 0.0, 0.0])
            table = torch.tensor(rows_local, dtype=torch.float64)         # one conversion, not one per slot
        for tid in self._place:
            self._trained[tid] += 1
        return table, (t0, t1, t2, n_local)

    def _gather(self, table: torch.Tensor, handle: Any = None) -> Tuple[torch.Tensor, List[Trial]]:
        """The ranks' tables stacked (read back to the host), and the trials of ranks that turned out to be gone.  With a
        ``handle`` the collective was posted earlier and is only waited for here."""
        t2 = time.time()
        lost: List[Trial] = []
        try:
            if handle is not None:
                gathered = handle.wait()
            else:
                gathered = self.comm.all_gather(table) if self.world > 1 else table.unsqueeze(0)
            gathered = gathered.cpu()
        except Exception as e:          # noqa: BLE001 - RCCL / gloo raise backend-specific types
            what = (f"rank {self.rank} of {self.world}: metric exchange after epoch {self.stats['epochs'] + 1} "
                    f"failed ({type(e).__name__}: {e})")
            if not self._repairable:
                raise RankLostError(what) from e
            lost = self._repair(what, e)
            try:        # the round's results of the trials that are still there, over the re-formed group
                gathered = self.comm.all_gather(table) if self.world > 1 else table.unsqueeze(0)
                gathered = gathered.cpu()
            except Exception as e2:     # noqa: BLE001
                raise RankLostError(f"{what}; the exchange over the re-formed group failed too "
                                    f"({type(e2).__name__}: {e2})") from e2
            # tables posted on the group that is gone are posted again, in order, on the new one
            self._inflight = [(self._post(tb), tb, tm) for _, tb, tm in self._inflight]
        self.stats["comm_s"] += max(0.0, time.time() - t2)
        return gathered, lost

    def _post(self, table: torch.Tensor) -> Any:
        return self.comm.all_gather_async(table) if self.world > 1 else _Ready(table.unsqueeze(0))

    def _discard_inflight(self) -> None:
        """Nothing is live any more: tables still in flight only hold rows of trials that were stopped meanwhile.  They
        are waited for (a posted collective must complete on every rank) and dropped."""
        while self._inflight:
            handle, _, _ = self._inflight.pop(0)
            try:
                handle.wait()
            except Exception:       # noqa: BLE001 - the search is over; a peer that already left is not an error
                pass

    def _async_round(self) -> List[Event]:
        """One round without a wait for the slowest rank in it: train and validate what is resident, post the table,
        free the slots of trials that have now trained ``max_epochs`` (their last results are in the table just posted),
        then read the table posted a round ago -- its collective has had a whole epoch to complete -- and hand its results
        to the runner.  Decisions therefore take effect one round late: a trial that a rung stops has trained one epoch
        more, whose result is dropped; in exchange no rank idles at a barrier every epoch."""
        posted = False
        if self._place and not self._must_read:
            table, times = self._train_round()
            self._inflight.append((self._post(table), table, times))
            posted = True
            for tid in [t for t in self._place if self._trained[t] >= self.t.max_epochs]:
                self._drain(self._live[tid])
        if not self._inflight:
            return []
        if self.exchange == "rungs":
            # read what is in flight only when something in it can be decided on (or nothing is left to train); then
            # read all of it, oldest first, one batch of events per call, before the next epoch is trained
            if not self._must_read:
                self._must_read = (not self._place) or self._decision_due()
            if not self._must_read:
                return []
        elif posted and len(self._inflight) < 2 and self._place:
            return []       # first round of a search: nothing is a round old yet
        handle, table, times = self._inflight.pop(0)
        if not self._inflight:
            self._must_read = False
        lost_trials, self._lost_pending = self._lost_pending, []
        gathered, lost = self._gather(table, handle)
        return self._report(gathered, times, lost_trials + lost)

    def _decision_due(self) -> bool:
        """Does the table just posted hold a result that can be answered with anything but CONTINUE?  Replicated: it looks
        at the iteration counts of the resident and the just-finished trials and asks the (replicated) scheduler."""
        ask = getattr(self.runner, "decides_at", None)
        if ask is None:
            return True
        its = {self._trained[tid] for tid in self._place} | {self._trained.get(tid, self.t.max_epochs) for tid in self._draining}
        return any(it >= self.t.max_epochs or ask(it) for it in its)

    def _drain(self, tr: Trial) -> None:
        """Asynchronous rounds: the trial has trained its last epoch; its slot goes back now, its last result follows."""
        owner, slot = self._place[tr.trial_id]
        if self.t.checkpoint_at_end or (self.runner is not None and self.runner.checkpoint_at_end):
            self._checkpoint(tr)
        self._place.pop(tr.trial_id)
        if owner == self.rank:
            self.engine.evict(slot)
        self._free[owner].append(slot)
        self._free[owner].sort()
        self._unload(tr.trial_id, owner)
        self._draining[tr.trial_id] = tr

    def _report(self, gathered: torch.Tensor, times: Tuple[float, float, float, int], lost_trials: List[Trial]) -> List[Event]:
        t0, t1, t2, n_local = times
        if getattr(self.engine, "device_table", None) is not None:      # the device's own clock split the round
            mine_row = gathered[self.rank, -1]
            total, train = float(mine_row[1]), float(mine_row[3])
            t1, t2 = t0 + train, t0 + total
        self.stats["epochs"] += 1
        self.stats["slot_epochs"] += n_local
        self.stats["train_s"] += t1 - t0
        self.stats["prepared_ahead"] = getattr(self.engine, "prepared_total", 0)
        if hasattr(self.engine, "stream_window"):
            self.stats["stream_window"] = str(self.engine.stream_window)
        self.stats["eval_s"] += t2 - t1

        by_index = {tr.index: tr for tr in self._live.values()}
        # numpy, not torch: a masked select + argsort of a few hundred rows costs tens of milliseconds per call through
        # the tensor library's intra-op thread pool and one millisecond here
        per_rank = gathered.reshape(self.world, self.capacity + 1, _REC)[:, -1, 1:3].tolist()
        self._slot_time = max((sec / n for sec, n in per_rank if n > 0), default=None)
        # the round's length as reported to schedulers and stop criteria: the slowest rank's, identical on every rank
        self._epoch_time = max(sec for sec, _ in per_rank)
        self.replicated_elapsed_s += self._epoch_time
        rows = gathered.reshape(-1, _REC).numpy()
        rows = rows[rows[:, 0] >= 0]
        rows = rows[np.argsort(rows[:, 0], kind="stable")]
        now = time.time()
        const = {"time_this_iter_s": self._epoch_time, "timestamp": int(now),
                 "date": time.strftime("%Y-%m-%d_%H-%M-%S", time.localtime(now)), "done": False,
                 "experiment_id": self.experiment_id, "pid": os.getpid(), "hostname": self._hostname,
                 "node_ip": "127.0.0.1"}
        events: List[Event] = []
        self._awaiting = {}
        self._n_undecided = 0
        on_nonfinite, isfinite, t_total, place = self.t.on_nonfinite, math.isfinite, self._t_total, self._place
        for idx, it, train_loss, val_loss, val_mse, val_mape in rows.tolist():
            tr = by_index.get(int(idx))
            if tr is None:
                continue        # asynchronous rounds: the extra epoch of a trial that was stopped while it trained
            tid = tr.trial_id
            t_total[tid] += self._epoch_time
            diverged = not (isfinite(train_loss) and isfinite(val_loss))
            if diverged and on_nonfinite == "error":
                # every rank sees the same table, so every rank frees the same slot
                self._drop(tr)
                self.stats["diverged"] += 1
                events.append(Event("error", tr, error=f"non-finite loss at epoch {int(it)}: "
                                                       f"train_loss={train_loss}, validation_loss={val_loss}"))
                continue
            rank, slot = place.get(tid) or (tr.placement["rank"], tr.placement["slot"])
            result = dict(const, training_iteration=int(it), train_loss=train_loss, validation_loss=val_loss,
                          validation_mse=val_mse, validation_mape=val_mape, time_total_s=t_total[tid], trial_id=tid,
                          iterations_since_restore=int(it), rank=rank, slot=slot)
            if diverged:
                result["diverged"] = True
                if on_nonfinite == "stop":
                    self._diverged.add(tid)
                    self.stats["diverged"] += 1
            self._awaiting[tid] = None
            self._n_undecided += 1
            events.append(Event("result", tr, result=result))
        # trials that sat on a rank that is gone: back to the runner, which queues them for a restart on the survivors
        events.extend(Event("lost", tr) for tr in lost_trials)
        return events

    # ------------------------------------------------------------------ a rank is gone
    def _repair(self, what: str, cause: Exception) -> List[Trial]:
        """Re-form the group from the surviving ranks and re-number everything that names a rank.  Returns the trials the
        lost ranks held.  Every survivor runs this with the same tables and (from ``shrink``) the same survivor list, so
        the replicated state stays replicated."""
        old_world = self.world
        try:
            survivors = self.comm.shrink()
        except Exception as e:          # noqa: BLE001 - no way to re-form the group: the whole job restarts
            raise RankLostError(f"{what}; the group could not be re-formed ({type(e).__name__}: {e})") from cause
        lost = sorted(set(range(old_world)) - set(survivors))
        new_of = {old: new for new, old in enumerate(survivors)}
        self.rank, self.world = self.comm.rank, self.comm.world_size
        self._free = [self._free[old] for old in survivors]
        self._load = [self._load[old] for old in survivors]
        lost_trials: List[Trial] = []
        for tid, (r, slot) in sorted(self._place.items()):
            tr = self._live[tid]
            if r in new_of:
                self._place[tid] = (new_of[r], slot)
                tr.placement = {"rank": new_of[r], "slot": slot}
            else:
                del self._place[tid]
                del self._live[tid]
                self._cost.pop(tid, None)
                self._trained.pop(tid, None)
                lost_trials.append(tr)
        lost_trials.sort(key=lambda t: t.index)
        trials = self.runner.trials if self.runner is not None else list(self._live.values()) + lost_trials
        for tr in trials:               # in-memory checkpoints are named by the rank that holds them
            src = getattr(tr, "checkpoint_rank", None)
            if src is not None:
                tr.checkpoint_rank = new_of.get(src)
                if tr.checkpoint_rank is None:
                    tr.checkpoint = None
                    if not tr.logdir:
                        # the only copy went with the lost rank and nothing is on disk: the trial starts over, and says
                        # so -- a restart from the initial weights under the old iteration count would pass for a resume
                        tr.checkpoint_iter = 0
        self.stats["world_size"] = self.world
        self.stats["ranks_lost"] = self.stats.get("ranks_lost", 0) + len(lost)
        self.stats["trials_recovered"] = self.stats.get("trials_recovered", 0) + len(lost_trials)
        if self.runner is not None:
            self.runner.on_ranks_lost(is_primary=(self.rank == 0))
        return lost_trials

    def _prefetch(self, t_issue: float) -> None:
        """Prepare upcoming trials while the issued epoch runs.  As many as are likely to get a slot when this epoch
        ends: the trials that reach ``max_epochs`` now, or as many as the previous round retired, whichever is more.
        That work has to be done before the next epoch anyway; done here it overlaps with the GPU instead of following
        it.  Only trials the runner can name in advance without changing the search (already pending, or drawn from a
        searcher that does not look at results)."""
        sure = sum(1 for tr in self._live.values() if tr.training_iteration + 1 >= self.t.max_epochs)
        # ... or, for an engine whose preparation is host arithmetic only (no capture, nothing on the device), as many as
        # the scheduler may stop when this epoch's results arrive: what is prepared too early waits in the engine's cache
        may_stop = 0
        if getattr(self.engine, "prepare_is_cheap", False):
            may_stop = sum(1 for tr in self._live.values() if self.runner.decides_at(tr.training_iteration + 1))
        want = min(self.capacity, max(sure, self._retired_last, may_stop))
        if want <= 0:
            return
        done = 0
        for tr in self.runner.lookahead(want):
            if tr.restore_from is not None or tr.start_time is not None:
                continue                              # resumes from a state: built when it gets its slot
            if not self.engine.prepare(tr.index, tr.config):
                break
            done += 1
            if done >= want:
                break

    def resolve(self, trial: Trial, decision: str) -> None:
        if trial.trial_id in self._diverged:      # a NaN/inf slot is retired whatever the scheduler said
            self._diverged.discard(trial.trial_id)
            decision = STOP
        if self._awaiting.get(trial.trial_id, "") is None:
            self._n_undecided -= 1
        self._awaiting[trial.trial_id] = decision
        if self._n_undecided == 0:
            self._flush()

    # ------------------------------------------------------------------ applying decisions
    def _flush(self) -> None:
        decisions, self._awaiting = self._awaiting, {}
        order = sorted(decisions, key=lambda tid: self._live[tid].index)
        # exploits first: a donor must still be resident when its weights are copied
        for tid in order:
            if decisions[tid] == RESTART:
                self._exploit(self._live[tid])
        for tid in order:
            tr, d = self._live[tid], decisions[tid]
            finished = d == STOP or tr.training_iteration >= self.t.max_epochs
            if finished:
                self._retire(tr)
            elif self.t.checkpoint_freq and tr.training_iteration % self.t.checkpoint_freq == 0 \
                    and tid in self._place:
                self._checkpoint(tr)
        self._retired_last, self._retired_now = self._retired_now, 0

    def _checkpoint(self, tr: Trial) -> None:
        owner, slot = self._place[tr.trial_id]
        # what the slot holds: the epochs it has trained, which under asynchronous rounds is one more than what has been
        # reported so far
        it = self._trained.get(tr.trial_id, tr.training_iteration) if self.async_rounds else tr.training_iteration
        tr.checkpoint_iter, tr.checkpoint_rank = it, owner      # known on every rank
        if owner != self.rank:
            tr.checkpoint = None          # an older state held here (the trial has moved since) is out of date now
            return
        state = self.engine.export(slot)
        state.setdefault("training_iteration", it)
        tr.checkpoint = state
        if tr.logdir:
            keep = self.runner.keep_checkpoints_num if self.runner is not None else None
            ckpt.save_checkpoint(tr.logdir, it, state, keep)

    def _unload(self, tid: str, rank: int) -> None:
        """Take a trial's weight off its rank (call after its slot went back to the free list)."""
        self._load[rank] -= self._cost.pop(tid, 0.0)
        if len(self._free[rank]) == self.capacity:
            self._load[rank] = 0.0          # no rounding residue on an empty rank

    def _drop(self, tr: Trial) -> None:
        self._trained.pop(tr.trial_id, None)
        if self._draining.pop(tr.trial_id, None) is not None:       # its slot went back when it finished training
            self._live.pop(tr.trial_id)
            return
        owner, slot = self._place.pop(tr.trial_id)
        if owner == self.rank:
            self.engine.evict(slot)
        self._free[owner].append(slot)
        self._free[owner].sort()
        self._unload(tr.trial_id, owner)
        self._live.pop(tr.trial_id)

    def _retire(self, tr: Trial) -> None:
        self._trained.pop(tr.trial_id, None)
        if self._draining.pop(tr.trial_id, None) is not None:       # its slot went back when it finished training
            self._live.pop(tr.trial_id)
            self._retired_now += 1
            self._queued.append(Event("done", tr, result={"decision": STOP}))
            return
        owner, slot = self._place.pop(tr.trial_id)
        if self.t.checkpoint_at_end or (self.runner is not None and self.runner.checkpoint_at_end):
            self._place[tr.trial_id] = (owner, slot)
            self._checkpoint(tr)
            self._place.pop(tr.trial_id)
        if owner == self.rank:
            self.engine.evict(slot)
        self._free[owner].append(slot)
        self._free[owner].sort()
        self._unload(tr.trial_id, owner)
        self._live.pop(tr.trial_id)
        self._retired_now += 1
        self._queued.append(Event("done", tr, result={"decision": STOP}))

    def _exploit(self, tr: Trial) -> None:
        req = getattr(tr, "pbt_request", None)
        tr.pbt_request = None
        if not req or req["donor"] not in self._place or tr.trial_id not in self._place:
            return
        (drank, dslot), (rrank, rslot) = self._place[req["donor"]], self._place[tr.trial_id]
        tr.config = req["config"]
        self.stats["exploits"] += 1
        if tr.logdir and rrank == self.rank:
            os.makedirs(tr.logdir, exist_ok=True)
            dump_json(tr.config, os.path.join(tr.logdir, "params.json"))
        if drank == rrank:
            if drank == self.rank:
                self.engine.clone(dslot, rslot, tr.config)
            return
        # donor and receiver live on different GPUs
        if hasattr(self.engine, "adopt_raw"):
            # the slot's device slab (parameters + both moments + counters) goes point-to-point: HBM -> xGMI -> HBM
            if self.rank == drank:
                self.comm.send(self.engine.export_raw(dslot), rrank)
            elif self.rank == rrank:
                buf = torch.empty(self.engine.raw_numel, dtype=torch.float32, device=self.device)
                self.engine.adopt_raw(rslot, self.comm.recv(buf, drank), tr.config)
            return
        # other engines ship the flattened state dict point-to-point
        from ..parallel.exchange import recv_state, send_state
        if self.rank == drank:
            send_state(self.comm, self.engine.export(dslot), rrank)
        elif self.rank == rrank:
            self.engine.import_state(rslot, recv_state(self.comm, self.engine.export(rslot), drank), tr.config)

    # ------------------------------------------------------------------ load balance between GPUs
    def _rebalance(self) -> None:
        """Move trials from the most loaded rank to the least loaded until no move is worth it.  Every rank holds the same placement tables, so all of them plan the same moves in the same
        order with no message exchanged; a move itself is one broadcast from the trial's old owner."""
        # A move has a price (building the trial again on the other GPU; for the slab engines next to nothing) and
        # buys one trial's time per round off the fullest rank.  It has to pay within the two rounds a trial that just
        # survived a rung can be expected to run at the least.  Both numbers are the same on every rank.
        cost = float(getattr(self.engine, "move_cost_s", 0.0))
        if cost > 0.0 and (self._slot_time is None or 2.0 * self._slot_time < cost):
            return
        t0, moved = time.time(), False
        gap = max(2, int(self.t.rebalance_gap))      # a difference of one cannot be improved, only passed back and forth
        # Weights, not head counts: the round lasts as long as the rank whose trials weigh the most.  A trial moves from
        # the heaviest rank to the lightest one that has a free slot; the trial taken is the one that brings the pair
        # closest to level (with equal weights: any, and the newest is taken).  It stops when the best move would gain less
        # than ``gap`` per cent of a round (equal weights: when the two ranks are fewer than ``gap`` trials apart).
        # Bounded: every move lowers the heavier of the two without lifting the lighter above it.
        for _ in range(4 * self.capacity):
            live = len(self._place)
            if live == 0 or self.world < 2:
                break
            unit = sum(self._load) / live
            hi = max(range(self.world), key=lambda r: (round(self._load[r], 12), -r))
            open_ = [r for r in range(self.world) if self._free[r] and r != hi]
            if not open_:
                break
            lo = min(open_, key=lambda r: (round(self._load[r], 12), r))
            diff = self._load[hi] - self._load[lo]
            cands = [(t, self._cost[t]) for t, (r, _) in self._place.items() if r == hi and self._cost[t] < diff * (1.0 - 1e-9)]
            if not cands:
                break
            tid, c = min(cands, key=lambda tc: (round(abs(tc[1] - diff / 2.0), 12), -self._live[tc[0]].index))
            costs = self._cost.values()
            if max(costs) - min(costs) <= 1e-12 * unit:
                if diff < gap * unit * (1.0 - 1e-9):      # equal weights: head counts, ``gap`` apart
                    break
            elif min(c, diff - c) < 0.01 * gap * sum(self._load) / self.world:
                break       # the best move would take less than gap per cent of a round off the heavier rank
            if not self._move(self._live[tid], lo):
                continue        # a rank vanished during the move and the group was re-formed: plan again for the survivors
            moved = True
        if moved:
            self.stats["migrate_s"] += time.time() - t0

    def _move(self, tr: Trial, dst: int) -> bool:
        """Move one trial to rank ``dst``; False if a rank was lost on the way (the group has been re-formed and every
        placement re-numbered: the caller plans again)."""
        src, sslot = self._place[tr.trial_id]
        dslot = self._free[dst][0]            # taken for good only once the state has arrived
        eng = self.engine
        err: Optional[Exception] = None
        payload = None
        try:
            if hasattr(eng, "export_raw"):
                # parameters and optimiser moments travel as the device-resident slab they already are: HBM -> xGMI -> HBM
                buf = eng.export_raw(sslot) if self.rank == src else \
                    torch.empty(eng.raw_numel, dtype=torch.float32, device=self.device)
                payload = self.comm.broadcast(buf, src=src)
            else:
                payload = self.comm.broadcast_object(eng.export(sslot) if self.rank == src else None, src=src)
        except Exception as e:          # noqa: BLE001 - RCCL / gloo raise backend-specific types
            err = e
        # A broadcast can complete on some ranks before the others notice that a peer is gone.  The move is committed only
        # when every rank says it holds the payload: one more (tiny) collective, which fails or reports 0 on every
        # survivor alike, so that either all of them commit the move or all of them repair the group and plan again.
        try:
            flag = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float32, device=self.comm.device)
            self.comm.all_reduce(flag, op="min")
            agreed = float(flag.item()) >= 1.0
        except Exception as e:          # noqa: BLE001
            agreed, err = False, (err or e)
        if not agreed:
            cause = err or RuntimeError("another rank did not receive the trial")
            what = (f"rank {self.rank} of {self.world}: moving trial {tr.trial_id} from rank {src} to "
                    f"rank {dst} failed ({type(cause).__name__}: {cause})")
            if not self._repairable:
                raise RankLostError(what) from cause
            self._lost_pending.extend(self._repair(what, cause))
            return False
        self._free[dst].pop(0)
        if self.rank == dst:
            if hasattr(eng, "import_raw"):
                eng.import_raw(dslot, tr.index, tr.config, payload)
            else:
                eng.load(dslot, tr.index, tr.config, state=payload)
        if self.rank == src:
            eng.evict(sslot)
        self._free[src].append(sslot)
        self._free[src].sort()
        self._place[tr.trial_id] = (dst, dslot)
        tr.placement = {"rank": dst, "slot": dslot}
        c = self._cost.get(tr.trial_id, 0.0)
        self._load[src] -= c
        self._load[dst] += c
        if len(self._free[src]) == self.capacity:
            self._load[src] = 0.0
        self.stats["migrations"] += 1
        return True

    def shutdown(self) -> None:
        self._discard_inflight()
        self.engine.close()
