"""GraphPackEngine: non-MLP trial families (FT-Transformer, custom factories) on the GPU, one captured hipGraph
per slot.

The grouped-GEMM engine packs MLP trials into shared launches; Transformer-style trials differ in depth, width and
head count, so they keep one module per slot -- but nothing else of the eager loop survives: parameters of a slot
live in one flat slab, the optimiser is the fused device-state kernel, and a slot's whole training step
(gather -> forward in bf16 autocast with the fused attention / LayerNorm kernels -> loss -> backward -> update) is
captured once and replayed for every batch of every epoch (``train/graphed.py``).  Row order stays the keyed
Feistel permutation of ``(seed, trial_index, epoch)``, so a trial's data order does not depend on its slot or GPU.

Slots do not wait for each other: one host thread issues the slots' graph replays round-robin over a few HIP
streams, so the small kernels of several trials are in flight at once (measured on one MI355X with the sample
Transformer, 25 steps x 4 trials: 65 ms replayed back to back on one stream, 21 ms on four streams; raising the
runtime's number of hardware queues above its default four made it slower, not faster).  Every slot owns its dropout counter, its
cursor and its loss accumulator, so the interleaving changes no trial's numbers.  Validation of all slots is
issued the same way and read back with one synchronisation.  In the small-kernel regime a slot's validation pass is
captured too and replayed right behind its last training step, so an epoch of the whole pack -- training and
validation of every slot -- is one stretch of asynchronous replays and ONE host synchronisation (eager validation cost
about 2 ms of host time per slot and epoch with the GPU idle meanwhile).

Concurrent issue is the default only in the regime where it pays: steps that touch at most ``SMALL_STEP_ROWS`` token
rows (launch-latency-bound kernels of a few workgroups each).  Packs with long token dimensions, whose GEMMs fill the
chip on their own, are issued on one stream, slot after slot, unless ``model_kwargs["streams"]`` asks otherwise.

Why big packs once stalled on several streams, and what makes them safe now.  Three bf16 FT-Transformer slots (66 560
token rows per step) replayed over three streams used to stop for good after a dozen steps: every stream idle, the host
blocked in a graph launch or in the final synchronisation.  ``bench/hang_repro.py`` reproduces it and bisects it
(``profiles/r2_multistream_stall.txt``): it happens with 2 or 3 slots, in bf16 and in f32, eagerly-built or replayed, with
the host held to 1, 4 or 16 replays ahead of each stream or not held at all, with 4 or 16 hardware queues -- and only
while the dense layers run on the BLAS library.  For these shapes the library picks "stream-K" GEMM kernels (``SK3`` in
their names), whose workgroups wait inside the launch for partial tiles of sibling workgroups and therefore need the
whole launch resident; two such launches from different streams split the CUs between them and each waits for
workgroups that the other keeps from being scheduled.  The small-kernel regime never met them (its GEMMs are plain
tiles).  Dense layers now run on this package's own tile kernels (``ops/linear.py``: no workgroup ever waits for
another), with which the same pack completes on any number of streams (156 steps x 3 slots in 1.8 s, unbounded
run-ahead); no library GEMM route is left in the package.  Independently of that, the host is kept at most ``OPTIONS.in_flight`` replays ahead of each stream it feeds (an event
ring per stream), which bounds the queue depth at no cost: in the small regime the device drains replays as fast as
one thread enqueues them.

PBT exploits inside a pack are slab-to-slab copies (weights + both moment slabs + step counter); the captured graph
of the receiving slot stays valid because it only holds pointers.
"""
from __future__ import annotations

import os
import threading
from typing import Dict, List, Optional

import numpy as np
import torch

from .engine import TorchPackEngine, _Slot
from .permutation import permute, round_keys
from ..models.mlp import MLPTrialSpec, init_mlp_state
from ..ops import rng
from ..options import OPTIONS
from ..train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
from ..train.losses import MAPE_EPS, elementwise_loss, get_criterion

DEFAULT_STREAMS = 4      # = the runtime's hardware queues per process; which streams exactly: packed/streams.py
SMALL_STEP_ROWS = 4096   # batch x tokens up to which slots are issued concurrently by default
# OPTIONS.in_flight: graph replays the host may be ahead of any one stream while it feeds several


class _FlowControl:
    """Bounds how far the host runs ahead of each of several streams: ``admit(j)`` before enqueueing work on stream
    ``j`` waits until all but the newest ``depth - 1`` pieces of work enqueued there have finished; ``sent(j, stream)``
    marks the piece just enqueued."""

    def __init__(self, n_streams: int, depth: Optional[int] = None):
        self.depth = max(1, int(OPTIONS.in_flight if depth is None else depth))
        self.rings: List[List[torch.cuda.Event]] = [[] for _ in range(n_streams)]
        self.count = [0] * n_streams

    def admit(self, j: int) -> None:
        k = self.count[j]
        if k >= self.depth:
            self.rings[j][k % self.depth].synchronize()

    def sent(self, j: int, stream: torch.cuda.Stream) -> None:
        k = self.count[j]
        ring = self.rings[j]
        if len(ring) < self.depth:
            ring.append(torch.cuda.Event())
        ring[k % self.depth].record(stream)
        self.count[j] = k + 1


class GraphPackEngine(TorchPackEngine):
    name = "graph"
    move_cost_s = 0.02       # a moved trial is built, uploaded and captured again on its new GPU
    _eval_warmed = threading.local()   # .done on threads where a validation pass has run once eagerly

    def __init__(self, trainable, device="cuda", capacity: Optional[int] = None, seed: int = 0):
        super().__init__(trainable, device, capacity, seed)
        if self.device.type != "cuda":
            raise ValueError("GraphPackEngine needs a GPU; use engine='torch' on the CPU")
        self._trainers: Dict[int, GraphedTrainer] = {}
        self._prepared: Dict[int, tuple] = {}          # trial_index -> (config, slot record, trainer), not yet in a slot
        self._pending_epoch = None
        self._prep_stream = torch.cuda.Stream(self.device)
        self._eval_ready = None        # slot -> [3] sums produced by the replays issued with the last training epoch
        self._eval_graphs = bool(int(trainable.model_kwargs.get("eval_graphs", OPTIONS.eval_graphs)))
        self.prepared_total = 0
        n = int(trainable.model_kwargs.get("streams", OPTIONS.graph_streams))
        self._n_streams = max(1, min(n, 16))
        # an explicit request for streams is honoured for every pack; the default keeps big-kernel packs on one stream
        self._streams_forced = "streams" in trainable.model_kwargs or OPTIONS.graph_force_streams
        self._streams: List[torch.cuda.Stream] = []     # picked at the first concurrent epoch (packed/streams.py)
        self._calibrated = False

    # ------------------------------------------------------------------ slot management
    def _make(self, trial_index, config, state=None):
        spec = self._spec(config)
        s = _Slot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        model, _ = self._build(spec, config, trial_index, with_optimizer=False)       # the fused flat optimiser below
        if state is None and self.t.model == "mlp":
            model.load_state_dict(init_mlp_state(spec, self.seed, trial_index))
        if state is not None and "model" in state:
            model.load_state_dict(state["model"])
        flat = FlatParams(model, grad_slab=False, bf16_shadow=self.amp, device=self.device)
        s.model = model
        s.opt = FlatOptimizer(flat, spec.optimizer, spec.learning_rate, spec.weight_decay, spec.momentum,
                              (spec.beta1, spec.beta2), spec.eps, spec.gradient_clipping, spec.warmup_steps,
                              spec.total_steps)
        s.epoch, s.step, s.train_loss = 0, 0, float("nan")
        if state is not None:
            if state.get("optimizer"):
                try:
                    s.opt.load_state_dict(state["optimizer"])
                except (ValueError, KeyError, TypeError):
                    pass       # another optimiser family or layout: fresh moments
            s.epoch, s.step = int(state.get("epoch", 0)), int(state.get("step", 0))
        merged = {**self.t.model_kwargs, **config}
        signature = (self.t.model if isinstance(self.t.model, str) else id(self.t.model), self.amp, spec.batch_size,
                     spec.optimizer, spec.loss_function, spec.gradient_clipping > 0, tuple(self.Xtr.shape[1:]),
                     tuple(tuple(p.shape) for p in flat.params),
                     tuple(sorted((k, v if isinstance(v, (str, bool)) else v > 0) for k, v in merged.items()
                                  if isinstance(v, (str, bool, int, float)))))
        trainer = GraphedTrainer(model, get_criterion(spec.loss_function), self.Xtr, self.ytr, spec.batch_size, s.opt,
                                 shuffle=False, drop_last=True, forward_ctx=self._autocast, signature=signature)
        self._capture_eval(s, trainer)
        return s, trainer

    # ------------------------------------------------------------------ validation
    def _eval_into(self, s: _Slot, acc: torch.Tensor) -> None:
        """Sums of the loss, the squared error and the absolute percentage error over the validation set -> acc[3]."""
        n = self.Xva.shape[0]
        was_training = s.model.training
        s.model.eval()
        with torch.no_grad():
            acc.zero_()
            for c0 in range(0, n, self.t.val_chunk):
                x, y = self.Xva[c0:c0 + self.t.val_chunk], self.yva[c0:c0 + self.t.val_chunk]
                with self._autocast():
                    o = s.model(x)
                o = o.float().reshape(y.shape)
                d = o - y
                acc += torch.stack([elementwise_loss(s.spec.loss_function, o, y).sum(), (d * d).sum(),
                                    (d.abs() / (y.abs() + MAPE_EPS)).sum()]).double()
        s.model.train(was_training)

    def _capture_eval(self, s: _Slot, trainer: GraphedTrainer) -> None:
        trainer.eval_acc = torch.zeros(3, device=self.device, dtype=torch.float64)
        trainer.eval_graph = None
        n = self.Xva.shape[0]
        small = min(n, self.t.val_chunk) * self._tokens_per_sample() <= 4 * SMALL_STEP_ROWS
        if self.amp or not small or not self._eval_graphs:
            return
        trainer.eval_graph = trainer.capture_callable(lambda: self._eval_into(s, trainer.eval_acc),
                                                      warm=not getattr(GraphPackEngine._eval_warmed, "done", False))
        GraphPackEngine._eval_warmed.done = True

    def prepare(self, trial_index, config) -> bool:
        """Build and capture a trial that has no slot yet (host work the executor overlaps with a running epoch);
        ``load`` then only files it.  Returns False when the cache is full."""
        if int(trial_index) in self._prepared:
            return True         # already waiting for its slot
        if len(self._prepared) >= self.capacity:
            return False
        # on a stream of its own: the host-to-device copies of a new model wait for the work queued before them on
        # their stream, and on the main stream that is the whole running epoch
        with torch.cuda.stream(self._prep_stream):
            self._prepared[int(trial_index)] = (dict(config),) + self._make(trial_index, config)
        self.prepared_total += 1
        return True

    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        ready = self._prepared.pop(int(trial_index), None)
        if ready is not None and state is None and ready[0] == dict(config):
            s, trainer = ready[1], ready[2]
        else:
            s, trainer = self._make(trial_index, config, state)
        self.slots[slot] = s
        self._trainers[slot] = trainer

    def evict(self, slot):
        self.slots.pop(slot, None)
        self._trainers.pop(slot, None)

    def export(self, slot):
        s = self.slots[slot]
        return {"model": {k: v.detach().cpu().clone() for k, v in s.model.state_dict().items()},
                "optimizer": s.opt.state_dict(), "epoch": s.epoch, "step": s.step, "trial_index": s.trial_index}

    def clone(self, src, dst, config):
        a, b = self.slots[src], self.slots[dst]
        if a.opt.flat.n != b.opt.flat.n or a.opt.name != self._spec(config).optimizer:
            return super().clone(src, dst, config)       # shapes differ: go through a state_dict
        with torch.no_grad():
            b.opt.flat.P.copy_(a.opt.flat.P)
            b.opt.flat.refresh_shadow()
            b.opt.M.copy_(a.opt.M)
            b.opt.V.copy_(a.opt.V)
            b.opt._i32[b.opt._STEP:b.opt._STEP + 1].copy_(a.opt._i32[a.opt._STEP:a.opt._STEP + 1])
            for (_, bb), (_, ab) in zip(b.model.named_buffers(), a.model.named_buffers()):
                bb.copy_(ab)
        b.spec, b.config = self._spec(config), dict(config)
        b.epoch, b.step = a.epoch, a.step
        self._apply_hparams(b)

    @staticmethod
    def _apply_hparams(s: _Slot) -> None:
        s.opt.set_hyper(s.spec.learning_rate, s.spec.weight_decay)

    # ------------------------------------------------------------------ training
    def _slot_seed(self, s: _Slot) -> int:
        # dropout draws of an epoch depend on (seed, trial, epoch) only, never on the slot or on neighbours
        return ((self.seed * 1_000_003 + s.trial_index) * 4099 + s.epoch) * 0x2545F491 + 77

    def _tokens_per_sample(self) -> int:
        X = self.Xtr
        return int(X.shape[1]) if X.dim() >= 2 else 1       # sequence length, or one token per tabular feature

    def _issue_streams(self, order, samples_per_launch: int, bound: int = SMALL_STEP_ROWS) -> List[torch.cuda.Stream]:
        """Streams the slots in ``order`` are spread over: several in the small-kernel regime, or when asked for."""
        small = samples_per_launch * self._tokens_per_sample() <= bound
        if not (small or self._streams_forced) or self._n_streams < 2 or len(order) < 2:
            return [torch.cuda.current_stream(self.device)]
        if not self._calibrated:
            self._calibrate_streams()
        return self._streams

    def _calibrate_streams(self) -> None:
        """Keep the window of consecutive pool streams on which independent kernel chains overlap best."""
        from .streams import pick_streams
        self._calibrated = True
        self._streams, self.stream_window = pick_streams(self.device, self._n_streams)

    def _fan_out(self, streams):
        cur = torch.cuda.current_stream(self.device)
        for st in streams:
            if st != cur:
                st.wait_stream(cur)
        return cur

    @staticmethod
    def _fan_in(cur, streams) -> None:
        for st in streams:
            if st != cur:
                cur.wait_stream(st)

    def train_epoch(self):
        self.train_epoch_begin()
        self.train_epoch_end()

    def train_epoch_begin(self):
        """Issue the epoch of every occupied slot; returns without waiting for the GPU."""
        order = self.occupied()
        self._pending_epoch = None
        if not order:
            return
        torch.cuda.current_stream(self.device).wait_stream(self._prep_stream)     # trials prepared ahead are complete
        n = self.Xtr.shape[0]
        plan = []
        for slot in order:
            s, tr = self.slots[slot], self._trainers[slot]
            steps = n // tr.B
            keys = round_keys(self.seed, s.trial_index, s.epoch)
            perm = permute(np.arange(steps * tr.B, dtype=np.uint32), n, keys).astype(np.int64)
            tr.perm[:steps * tr.B].copy_(torch.from_numpy(perm))      # blocking: the host array dies with this frame
            tr.seed(self._slot_seed(s))
            plan.append((s, tr, tr.begin_epoch()))
        streams = self._issue_streams(order, max(tr.B for _, tr, _ in plan))
        cur = self._fan_out(streams)
        ns = len(streams)
        ride = all(tr.eval_graph is not None for _, tr, _ in plan)     # validation rides behind the last step
        if ns == 1:                                   # slot after slot: a slot's working set stays in cache
            for _, tr, batches in plan:
                for i in range(batches):
                    tr.step_once(i)
                if ride:
                    tr.eval_graph.replay()
        else:
            flow = _FlowControl(ns)
            for i in range(max(b for _, _, b in plan) + 1):
                for j, (_, tr, batches) in enumerate(plan):
                    if i > batches or (i == batches and not ride):
                        continue
                    q = j % ns
                    flow.admit(q)                       # never more than OPTIONS.in_flight replays ahead of a stream
                    with torch.cuda.stream(streams[q]):
                        if i < batches:
                            tr.step_once(i)
                        else:
                            tr.eval_graph.replay()
                    flow.sent(q, streams[q])
        self._fan_in(cur, streams)
        parts = [torch.stack([tr.epoch_loss(b) for _, tr, b in plan]).double()]
        if ride:
            parts.append(torch.stack([tr.eval_acc for _, tr, _ in plan]).reshape(-1))
        self._pending_epoch = (plan, torch.cat(parts), ride, order)

    def train_epoch_end(self):
        """Wait for the issued epoch and book its losses."""
        if self._pending_epoch is None:
            return
        plan, stacked, ride, order = self._pending_epoch
        self._pending_epoch = None
        vals = stacked.tolist()                                                      # one synchronisation
        for (s, tr, batches), loss in zip(plan, vals):
            s.train_loss = loss
            s.step += batches
            s.epoch += 1
        k = len(plan)
        self._eval_ready = {slot: vals[k + 3 * j:k + 3 * j + 3] for j, slot in enumerate(order)} if ride else None

    @torch.no_grad()
    def evaluate(self):
        order = self.occupied()
        if not order:
            return {}
        n = self.Xva.shape[0]
        ready, self._eval_ready = self._eval_ready, None
        if ready is not None and sorted(ready) == order:        # produced by the replays behind the last training epoch
            rows = [ready[slot] for slot in order]
        else:
            acc = torch.zeros(len(order), 3, device=self.device, dtype=torch.float64)  # loss, squared error, APE sums
            streams = self._issue_streams(order, min(n, self.t.val_chunk), 4 * SMALL_STEP_ROWS)
            cur = self._fan_out(streams)
            ns = len(streams)
            flow = _FlowControl(ns) if ns > 1 else None
            for j, slot in enumerate(order):
                tr = self._trainers[slot]
                if flow is not None:
                    flow.admit(j % ns)
                with torch.cuda.stream(streams[j % ns]):
                    if tr.eval_graph is not None:
                        tr.eval_graph.replay()
                        acc[j].copy_(tr.eval_acc)
                    else:
                        self._eval_into(self.slots[slot], acc[j])
                if flow is not None:
                    flow.sent(j % ns, streams[j % ns])
            self._fan_in(cur, streams)
            rows = acc.tolist()                                                        # one synchronisation
        return {slot: {"train_loss": self.slots[slot].train_loss, "validation_loss": r[0] / n,
                       "validation_mse": r[1] / n, "validation_mape": 100.0 * r[2] / n}
                for slot, r in zip(order, rows)}
