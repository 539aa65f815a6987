"""Data-parallel training of one oversized trial across all GPUs of the node.

The reference only ever trains a model on a single device (/root/reference/ray-tune-hpo-regression.py:286-287).
A trial that does not fit the packed regime -- a width-8192 MLP on a 100M x 256 table -- is instead run
data-parallel: every rank holds a shard of the table in its own HBM, trains on its slice of the global
batch, and gradients are averaged with RCCL all-reduce over xGMI.

MI355X-first choices:

* parameters, gradients and optimiser moments are FLAT slabs.  On a GPU (``HipDPEngine``) a whole optimiser step is
  enqueued by the C++ driver ``ops/csrc/dp.hip``: keyed gather of bf16 rows, every GEMM on this package's bf16 MFMA
  tile kernels (forward + bias + ReLU, split-K weight gradient + bias gradient, input gradient + ReLU mask), fused
  head + loss, per-layer bf16 gradient buckets, and the fused optimiser that consumes bf16 gradients, updates the f32
  masters and refreshes the bf16 shadow in one pass -- no autograd, no library GEMM, no Python between kernels;
* gradients are reduced in BUCKETS, one per layer: the moment a layer's weight gradient is in its bucket, the bucket's
  ``ncclAllReduce`` is enqueued on a communication stream (a direct RCCL communicator, ``parallel/rccl.py``), so the
  all-reduce of layer l overlaps the input-gradient GEMM of layer l and the whole backward of layers < l; the
  optimiser waits on one event.  On the CPU (tests, gloo) the same schedule runs as autograd hooks + asynchronous
  ``torch.distributed`` all-reduces.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce moves
  2 * (n-1)/n * bytes per link, i.e. ~1.5 ms for one 8192 x 8192 bf16 bucket -- about one layer's backward,
  which is why per-layer buckets (not one big late bucket, not many tiny ones) are the right size here;
* bf16 buckets halve the bytes on the links compared with f32 gradients;
* ``dtype="f32"`` (config key or argument) selects the exact-f32 form of the same driver: f32 table, activations and
  buckets, every GEMM on the exact-f32 tile kernels, gradient clipping inside the optimiser launch.  Its results do not
  depend on how many ranks share the batch beyond the order of the all-reduce's additions, and on the CPU path (gloo)
  they are bitwise equal between 1, 2 and 4 ranks' worth of identical shards.

A GPU trial always runs on the C++ driver: shapes it cannot take (a feature count or a width that is not a multiple
of 8, more than 8 hidden layers, clipping with bf16 buckets) raise ``ValueError`` with the remedy instead of falling back
to an eager loop.  The autograd implementation below is the CPU path (tests, gloo rehearsals) and the oracle the GPU
tests compare with.

Every rank runs the same HPO driver (same seed, same searcher, same scheduler): trials execute one
after another, each using all ranks, and the reported metrics are all-reduced so the replicated
schedulers take identical decisions.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .comm import Comm, LocalComm
from ..options import OPTIONS
from ..models.mlp import MLPTrialSpec, init_mlp_state, lr_factor
from ..packed.permutation import permute, round_keys
from ..train.losses import MAPE_EPS, elementwise_loss
from ..train.optim import OPT_CODES, build_optimizer


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class FlatMLP:
    """MLP whose weights, gradients and optimiser state live in flat slabs (bucket = one layer)."""

    def __init__(self, spec: MLPTrialSpec, device: torch.device, seed: int, trial_index: int, low_precision: bool,
                 autograd: bool = True):
        self.spec, self.device, self.low = spec, device, low_precision
        dims = spec.layer_dims
        self.offsets: List[Tuple[int, int, int, int]] = []     # (w_off, b_off, out, in) per layer incl. head
        off = 0
        self.bucket_bounds: List[Tuple[int, int]] = []
        for o, i in dims:
            start = off
            w_off = off
            off = _round_up(off + o * i, 8)
            b_off = off
            off = _round_up(off + o, 8)
            self.offsets.append((w_off, b_off, o, i))
            self.bucket_bounds.append((start, off))
        self.numel = off
        self.P = torch.zeros(off, device=device, dtype=torch.float32)
        state = init_mlp_state(spec, seed, trial_index)
        names = [f"layers.{l}" for l in range(len(dims) - 1)] + ["head"]
        for (w_off, b_off, o, i), name in zip(self.offsets, names):
            self.P[w_off:w_off + o * i] = state[f"{name}.weight"].reshape(-1).to(device)
            self.P[b_off:b_off + o] = state[f"{name}.bias"].to(device)
        wdtype = torch.bfloat16 if low_precision else torch.float32
        self.W = self.P.to(wdtype) if low_precision else self.P      # compute weights (bf16 shadow on GPU)
        self.params: List[torch.Tensor] = []
        if not autograd:        # the GPU engine has its own gradient buffers and never builds an autograd graph
            self.Gf = None
            return
        self.Gf = torch.zeros(off, device=device, dtype=wdtype)      # flat gradients
        for w_off, b_off, o, i in self.offsets:
            w = self.W[w_off:w_off + o * i].view(o, i).requires_grad_(True)
            b = self.W[b_off:b_off + o].requires_grad_(True)
            w.grad = self.Gf[w_off:w_off + o * i].view(o, i)
            b.grad = self.Gf[b_off:b_off + o]
            self.params += [w, b]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The CPU path's forward (plain PyTorch: the oracle of the GPU driver)."""
        n = len(self.offsets)
        for l in range(n):
            x = F.linear(x, self.params[2 * l], self.params[2 * l + 1])
            if l < n - 1:
                x = torch.relu(x)
        return x.squeeze(-1)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        names = [f"layers.{l}" for l in range(len(self.offsets) - 1)] + ["head"]
        out = {}
        for (w_off, b_off, o, i), name in zip(self.offsets, names):
            out[f"{name}.weight"] = self.P[w_off:w_off + o * i].view(o, i).detach().cpu().clone()
            out[f"{name}.bias"] = self.P[b_off:b_off + o].detach().cpu().clone()
        return out


class BucketedAllReduce:
    """Launch one asynchronous all-reduce per gradient bucket as soon as the bucket is complete."""

    def __init__(self, model: FlatMLP, comm: Comm):
        self.m, self.comm = model, comm
        self.pending: List[Any] = []
        self._ready = [0] * len(model.bucket_bounds)
        self.bytes_reduced = 0
        if comm.world_size > 1 and model.params:
            for l in range(len(model.offsets)):
                for p in (model.params[2 * l], model.params[2 * l + 1]):
                    p.register_post_accumulate_grad_hook(self._make_hook(l))

    def _make_hook(self, layer: int) -> Callable:
        def hook(_param):
            self._ready[layer] += 1
            if self._ready[layer] == 2:          # weight and bias of this layer are both in the slab
                lo, hi = self.m.bucket_bounds[layer]
                bucket = self.m.Gf[lo:hi]
                self.pending.append(self.comm.all_reduce(bucket, op="avg", async_op=True))
                self.bytes_reduced += bucket.numel() * bucket.element_size()
        return hook

    def wait(self) -> None:
        for w in self.pending:
            if w is not None:
                w.wait()
        self.pending.clear()
        self._ready = [0] * len(self._ready)


class DPTrainer:
    """One trial, all ranks.  ``X``/``y`` are this rank's shard (rows are disjoint across ranks)."""

    def __init__(self, config: Dict, X: torch.Tensor, y: torch.Tensor, Xval: torch.Tensor, yval: torch.Tensor,
                 comm: Optional[Comm] = None, seed: int = 0, trial_index: int = 0, dtype: Optional[str] = None):
        self.comm = comm or LocalComm(X.device)
        self.device = X.device
        self.spec = MLPTrialSpec.from_config(config, int(X.shape[1]))
        self.gpu = self.device.type == "cuda"
        self.dtype = str(config.get("dtype", dtype) or ("bf16" if self.gpu else "f32"))
        if self.dtype not in ("bf16", "f32") or (self.dtype == "bf16" and not self.gpu):
            raise ValueError(f"dtype {self.dtype!r}: a data-parallel trial computes in 'bf16' (GPU only) or 'f32'")
        self.X, self.y, self.Xval, self.yval = X, y, Xval, yval
        if self.gpu:        # bf16: the table is kept in bf16 -- half the HBM, and it is what the GEMMs consume
            want = torch.bfloat16 if self.dtype == "bf16" else torch.float32
            self.X = X if X.dtype == want else X.to(want)
            self.Xval = Xval if Xval.dtype == want else Xval.to(want)
            self._check_gpu_shapes()
        self.model = FlatMLP(self.spec, self.device, seed, trial_index, low_precision=self.gpu and self.dtype == "bf16",
                             autograd=not self.gpu)
        self.reducer = BucketedAllReduce(self.model, self.comm)
        self.seed, self.trial_index = seed, trial_index
        self.global_batch = self.spec.batch_size
        if self.global_batch % self.comm.world_size:
            raise ValueError("batch_size must be divisible by the number of ranks")
        self.local_batch = self.global_batch // self.comm.world_size
        # Shards differ by a row (shard_rows gives +-1), and every rank must take the same number of steps -- each step is
        # a fixed sequence of collectives.  The step count comes from the smallest shard, agreed on once.
        n_min = torch.tensor([int(X.shape[0])], dtype=torch.int64, device=self.device if self.gpu else "cpu")
        if self.comm.world_size > 1:
            self.comm.all_reduce(n_min, op="min")
        self.steps_per_epoch = int(n_min.item()) // self.local_batch
        self.epoch, self.step = 0, 0
        self.stats = {"steps": 0, "train_s": 0.0}
        self.engine = None
        if self.gpu:
            self.engine = HipDPEngine(self)
            self.M, self.V = self.engine.M, self.engine.V
        else:
            self.opt = build_optimizer(self.spec.optimizer, [self.model.P], self.spec.learning_rate,
                                       self.spec.weight_decay, self.spec.momentum, self.spec.beta1, self.spec.beta2,
                                       self.spec.eps)

    def _check_gpu_shapes(self) -> None:
        """What ``ops/csrc/dp.hip`` takes; anything else is an error with its remedy, never a silent slower path."""
        F_, hidden = int(self.X.shape[1]), list(self.spec.hidden_sizes)
        if F_ % 8:
            raise ValueError(f"the table has {F_} features: the data-parallel engine reads rows in 16-byte pieces, pad the "
                             f"table with {(-F_) % 8} zero column(s) to a multiple of 8")
        if F_ // (8 if self.dtype == "bf16" else 4) > 256:
            raise ValueError(f"{F_} features exceed the row gather of the data-parallel engine "
                             f"({2048 if self.dtype == 'bf16' else 1024} in {self.dtype})")
        if any(h % 8 for h in hidden) or not 1 <= len(hidden) <= 8:
            raise ValueError(f"hidden_sizes {hidden}: the data-parallel engine takes 1 to 8 hidden layers whose widths are "
                             "multiples of 8")
        if self.spec.gradient_clipping > 0 and self.dtype == "bf16":
            raise ValueError("gradient_clipping with bf16 gradient buckets is not supported (the norm would be taken of "
                             "rounded gradients): use dtype='f32', which clips inside the optimiser launch")

    def close(self) -> None:
        """Destroy what the trial created outside the allocator: the direct RCCL communicator of the GPU engine."""
        if self.engine is not None:
            self.engine.close()

    # ------------------------------------------------------------------ optimiser (CPU path)
    def _apply_update(self) -> None:
        sp = self.spec
        for g in self.opt.param_groups:
            g["lr"] = sp.learning_rate * lr_factor(self.step, sp.warmup_steps, sp.total_steps)
        self.model.P.grad = self.model.Gf
        self.opt.step()

    # ------------------------------------------------------------------ epoch
    def train_epoch(self) -> float:
        t0 = time.time()
        n_local = int(self.X.shape[0])
        steps = self.steps_per_epoch
        if self.engine is not None:
            mean = self.engine.train_epoch(steps)
            self.reducer.bytes_reduced = self.engine.bytes_reduced
            self.step += steps
            self.epoch += 1
            self.stats["steps"] += steps
            self.stats["train_s"] += time.time() - t0
            if self.comm.world_size > 1:
                self.comm.all_reduce(mean, op="avg")
            return float(mean)
        keys = round_keys(self.seed, self.trial_index * 1009 + self.comm.rank, self.epoch)
        perm = torch.from_numpy(permute(np.arange(steps * self.local_batch, dtype=np.uint32), n_local, keys)).to(self.device)
        loss_sum = torch.zeros((), device=self.device, dtype=torch.float32)
        clip = self.spec.gradient_clipping
        for i in range(steps):
            idx = perm[i * self.local_batch:(i + 1) * self.local_batch]
            xb, yb = self.X.index_select(0, idx), self.y.index_select(0, idx)
            self.model.Gf.zero_()
            out = self.model.forward(xb).float()
            loss = elementwise_loss(self.spec.loss_function, out, yb).mean()
            loss.backward()                       # hooks launch the per-layer all-reduces as gradients appear
            self.reducer.wait()
            if clip > 0:                          # global norm of the averaged gradient, identical on all ranks
                norm = self.model.Gf.float().norm()
                self.model.Gf.mul_(torch.clamp(clip / (norm + 1e-6), max=1.0).to(self.model.Gf.dtype))
            self._apply_update()
            self.step += 1
            loss_sum += loss.detach()
        self.epoch += 1
        self.stats["steps"] += steps
        self.stats["train_s"] += time.time() - t0
        mean = loss_sum / max(steps, 1)
        if self.comm.world_size > 1:
            self.comm.all_reduce(mean, op="avg")
        return float(mean)

    @torch.no_grad()
    def evaluate(self, chunk: int = 65536) -> Dict[str, float]:
        """Validation metrics over the union of all ranks' validation shards (sample-weighted)."""
        acc = torch.zeros(4, device=self.device, dtype=torch.float64)
        if self.engine is not None:
            acc[:3] = self.engine.evaluate()
            acc[3] = float(self.Xval.shape[0])
            if self.comm.world_size > 1:
                self.comm.all_reduce(acc, op="sum")
            n = max(float(acc[3]), 1.0)
            return {"validation_loss": float(acc[0]) / n, "validation_mse": float(acc[1]) / n,
                    "validation_mape": 100.0 * float(acc[2]) / n}
        for c0 in range(0, int(self.Xval.shape[0]), chunk):
            x, y = self.Xval[c0:c0 + chunk], self.yval[c0:c0 + chunk]
            o = self.model.forward(x).float()
            d = o - y
            acc[0] += elementwise_loss(self.spec.loss_function, o, y).sum().double()
            acc[1] += (d * d).sum().double()
            acc[2] += (d.abs() / (y.abs() + MAPE_EPS)).sum().double()
            acc[3] += y.numel()
        if self.comm.world_size > 1:
            self.comm.all_reduce(acc, op="sum")
        n = max(float(acc[3]), 1.0)
        return {"validation_loss": float(acc[0]) / n, "validation_mse": float(acc[1]) / n,
                "validation_mape": 100.0 * float(acc[2]) / n}


class HipDPEngine:
    """The GPU side of a ``DPTrainer``: slabs, activation buffers, descriptor tables and the plan that
    ``ops/csrc/dp.hip`` walks.  It shares the f32 master slab ``P`` with the trainer's ``FlatMLP`` (so ``state_dict``
    keeps working) and takes over its bf16 shadow."""

    EVAL_CHUNK = 16384

    def __init__(self, tr: "DPTrainer"):
        from ..ops import lib as L
        from ..packed.permutation import half_bits
        from ..train.losses import LOSS_CODES
        self.tr, self.L, self.lib = tr, L, L.load()
        dev, m, sp = tr.device, tr.model, tr.spec
        self.device = dev
        self.f32 = tr.dtype == "f32"
        act = torch.float32 if self.f32 else torch.bfloat16
        hidden = list(sp.hidden_sizes)
        nl = len(hidden)
        self.F = int(tr.X.shape[1])
        b = tr.local_batch
        rows_max = max(b, min(self.EVAL_CHUNK, max(int(tr.Xval.shape[0]), 1)))
        self.eval_chunk = rows_max
        n = _round_up(m.numel, 8)
        self.numel = n
        # wgrad splits per layer: enough workgroups to fill the chip when a layer has few 128x128 tiles
        ks = []
        dims_in = [self.F] + hidden[:-1]
        edge, rows_per_split = (64, 128) if self.f32 else (128, 64)       # tile edge of the weight-gradient kernel in use
        for h, k in zip(hidden, dims_in):
            tiles = ((h + edge - 1) // edge) * ((k + edge - 1) // edge)
            ks.append(max(1, min(8, (b + rows_per_split - 1) // rows_per_split, (512 + tiles - 1) // tiles)))
        self.P, self.W = m.P, m.W
        if n != m.numel:
            raise ValueError("flat slab length must be a multiple of 8")      # FlatMLP rounds every tensor up to 8
        self.G32 = torch.zeros(max(ks) * n, device=dev, dtype=torch.float32)
        self.Gb = torch.zeros(n, device=dev, dtype=act)          # the buckets: what is all-reduced, what the optimiser reads
        self.M, self.V = torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.X0 = torch.zeros(b * self.F, device=dev, dtype=act)
        self.ybuf = torch.zeros(b, device=dev, dtype=torch.float32)
        self.H = [torch.zeros(rows_max * h, device=dev, dtype=act) for h in hidden]
        self.dZ = [torch.zeros(b * h, device=dev, dtype=act) for h in hidden]
        self.loss_acc = torch.zeros(1, device=dev, dtype=torch.float32)
        self.metrics = torch.zeros(3, device=dev, dtype=torch.float64)
        self.ids = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(int(self.lib.mt_head_scratch_floats(1, rows_max, max(hidden))), device=dev)
        hw_off, hb_off = m.offsets[nl][0], m.offsets[nl][1]

        def head_table(rows, y_ptr):
            t = np.zeros(1, dtype=L.HEAD_DTYPE)
            t[0] = (self.H[-1].data_ptr(), self.dZ[-1].data_ptr(), self.P.data_ptr() + hw_off * 4,
                    self.P.data_ptr() + hb_off * 4, self.G32.data_ptr() + hw_off * 4, self.G32.data_ptr() + hb_off * 4, y_ptr,
                    self.loss_acc.data_ptr(), self.metrics.data_ptr(), rows, hidden[-1], hidden[-1],
                    LOSS_CODES[sp.loss_function], 1 << 30, 0)
            return torch.from_numpy(t.view(np.uint8).reshape(-1).copy()).to(dev)

        self._head_train = head_table(b, self.ybuf.data_ptr())
        self._head_eval = head_table(rows_max, tr.yval.data_ptr())
        hp = np.zeros(1, dtype=L.OPT_DTYPE)
        clip = float(sp.gradient_clipping) if self.f32 else 0.0
        hp[0] = (sp.learning_rate, sp.beta1, sp.beta2, sp.eps, sp.weight_decay, sp.momentum, clip, OPT_CODES[sp.optimizer],
                 sp.warmup_steps, sp.total_steps, m.numel, 1 << 30, 0, 0, 0, 0)
        self._opt = torch.from_numpy(hp.view(np.uint8).reshape(-1).copy()).to(dev)
        self._clip_blocks = (n + 2047) // 2048 if clip > 0 else 0              # one partial norm per optimiser workgroup
        self._clip_partials = torch.zeros(max(self._clip_blocks, 1), device=dev, dtype=torch.float32)
        # collectives: a direct RCCL communicator when there is more than one rank
        self.rccl = None
        # OPTIONS.dp_force_collectives issues the per-bucket all-reduces even with one rank (over a one-rank communicator
        # they are copies): the event / stream choreography of the C++ driver can then be exercised on a single GPU
        force = OPTIONS.dp_force_collectives
        if tr.comm.world_size > 1 or force:
            from .rccl import RcclComm
            self.rccl = RcclComm(tr.comm, dev)
        self._events = [torch.cuda.Event() for _ in range(L.MT_MAX_LAYERS + 2)]
        for e in self._events:
            e.record()                        # creates the underlying HIP events
        plan = L.DpPlan()
        plan.n_hidden, plan.batch, plan.F = nl, b, self.F
        plan.n_rows, plan.hb = int(tr.X.shape[0]), half_bits(int(tr.X.shape[0]))
        plan.world = max(tr.comm.world_size, 2 if force else 1)
        plan.f32, plan.use_clip, plan.clip_blocks = int(self.f32), int(clip > 0), self._clip_blocks
        plan.clip_partials = self._clip_partials.data_ptr()
        for l, (h, k) in enumerate(zip(hidden, ks)):
            plan.widths[l], plan.ksplits[l] = h, k
            plan.H[l], plan.dZ[l] = self.H[l].data_ptr(), self.dZ[l].data_ptr()
        for l in range(nl + 1):
            plan.w_off[l], plan.b_off[l] = m.offsets[l][0], m.offsets[l][1]
            plan.bucket_lo[l], plan.bucket_hi[l] = m.bucket_bounds[l]
        plan.numel = n
        plan.X, plan.Y = tr.X.data_ptr(), tr.y.data_ptr()
        plan.X0, plan.ybuf = self.X0.data_ptr(), self.ybuf.data_ptr()
        plan.P, plan.W, plan.G32, plan.Gb = self.P.data_ptr(), self.W.data_ptr(), self.G32.data_ptr(), self.Gb.data_ptr()
        # the all-reduce is in place, except in the forced one-rank rehearsal: RCCL turns a one-rank in-place all-reduce
        # into nothing at all, an out-of-place one into a device copy on the communication stream -- bytes that really move
        # while the next input gradient runs, which is what a profile of the overlap needs to show
        self.Gr = torch.zeros_like(self.Gb) if (force and tr.comm.world_size == 1) else self.Gb
        plan.Gr = self.Gr.data_ptr()
        plan.M, plan.V = self.M.data_ptr(), self.V.data_ptr()
        plan.head, plan.opt, plan.ids = self._head_train.data_ptr(), self._opt.data_ptr(), self.ids.data_ptr()
        plan.head_scratch = self.scratch.data_ptr()
        if self.rccl is not None:
            from .rccl import DTYPES, OPS
            plan.all_reduce, plan.comm = self.rccl.fn_all_reduce, self.rccl.comm_handle
            plan.nccl_bf16, plan.nccl_avg, plan.nccl_f32 = DTYPES[torch.bfloat16], OPS["avg"], DTYPES[torch.float32]
            plan.comm_stream = self.rccl.stream.cuda_stream
        for l in range(nl + 1):
            plan.ready[l] = self._events[l].cuda_event
        plan.reduced = self._events[-1].cuda_event
        self.plan = plan
        self.plan_eval = L.DpPlan.from_buffer_copy(plan)
        self.plan_eval.head = self._head_eval.data_ptr()
        self.bytes_reduced = 0

    def train_epoch(self, steps: int) -> torch.Tensor:
        tr = self.tr
        keys = np.asarray(round_keys(tr.seed, tr.trial_index * 1009 + tr.comm.rank, tr.epoch), dtype=np.uint32)
        self.loss_acc.zero_()
        stream = torch.cuda.current_stream(self.device)
        if self.rccl is not None:
            self.rccl.stream.wait_stream(stream)
        self.L.check(self.lib.mt_dp_train_steps(C.byref(self.plan), keys.ctypes.data, 0, steps, tr.step, stream.cuda_stream),
                     "mt_dp_train_steps")
        if self.rccl is not None:
            self.bytes_reduced += steps * tr.model.numel * (4 if self.f32 else 2)
        return self.loss_acc[0] / max(steps, 1)

    def close(self) -> None:
        if self.rccl is not None:
            torch.cuda.synchronize(self.device)
            self.rccl.close()
            self.rccl = None
            self.plan.all_reduce = self.plan_eval.all_reduce = None
            self.plan.world = self.plan_eval.world = 1

    def evaluate(self) -> torch.Tensor:
        self.metrics.zero_()
        n = int(self.tr.Xval.shape[0])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.L.check(self.lib.mt_dp_eval(C.byref(self.plan_eval), self.tr.Xval.data_ptr(), n, self.eval_chunk, stream), "mt_dp_eval")
        return self.metrics.clone()


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, near-equal row range of ``rank``."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def dp_mlp_trainable(config: Dict, train: Tuple[torch.Tensor, torch.Tensor], val: Tuple[torch.Tensor, torch.Tensor],
                     comm: Optional[Comm] = None, epochs: int = 4, seed: int = 0) -> None:
    """Function trainable (SPMD): call under ``mitune.run`` on every rank with this rank's data shards."""
    from ..tune import session
    tid = session.get_trial_id() or "0"
    digits = "".join(ch for ch in tid if ch.isdigit())
    trainer = DPTrainer(config, train[0], train[1], val[0], val[1], comm=comm, seed=seed,
                        trial_index=int(digits[-6:]) if digits else 0)
    if "steps_per_epoch" in config:           # a report every so many steps instead of once per pass over the table
        trainer.steps_per_epoch = max(1, min(int(config["steps_per_epoch"]), trainer.steps_per_epoch))
    try:
        for _ in range(int(config.get("epochs", epochs))):
            train_loss = trainer.train_epoch()
            metrics = trainer.evaluate()
            metrics["train_loss"] = train_loss
            metrics["allreduce_gb"] = trainer.reducer.bytes_reduced / 1e9
            metrics["steps"] = trainer.step
            session.report(metrics)
    finally:
        trainer.close()
