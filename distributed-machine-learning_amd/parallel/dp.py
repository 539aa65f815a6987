"""Data-parallel training of one oversized trial across all GPUs of the node.

The reference only ever trains a model on a single device (/root/reference/ray-tune-hpo-regression.py:286-287).
A trial that does not fit the packed regime -- a width-8192 MLP on a 100M x 256 table -- is instead run
data-parallel: every rank holds a shard of the table in its own HBM, trains on its slice of the global
batch, and gradients are averaged with RCCL all-reduce over xGMI.

MI355X-first choices:

* parameters, gradients and optimiser moments are FLAT slabs; the model's tensors are views into them.
  On a GPU the weights the GEMMs read are the bf16 shadow slab, gradients are produced in bf16 straight
  into the flat gradient slab, and the fused HIP optimiser (``ops/csrc/optim.hip``) consumes bf16
  gradients, updates the f32 master slab and refreshes the bf16 shadow in one pass;
* gradients are reduced in BUCKETS, one per layer, each launched asynchronously from an autograd hook
  the moment that layer's gradients exist, so the all-reduce of layer l overlaps the backward GEMMs of
  layers < l.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce moves
  2 * (n-1)/n * bytes per link, i.e. ~1.5 ms for one 8192 x 8192 bf16 bucket -- about one layer's backward,
  which is why per-layer buckets (not one big late bucket, not many tiny ones) are the right size here;
* bf16 buckets halve the bytes on the links compared with f32 gradients.

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
from ..models.mlp import MLPTrialSpec, init_mlp_state, lr_factor
from ..packed.permutation import permute, round_keys
from ..train.losses import MAPE_EPS, elementwise_loss
from ..train.optim import OPT_CODES, build_optimizer


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class FlatMLP:
    """MLP whose weights, gradients and optimiser state live in flat slabs (bucket = one layer)."""

    def __init__(self, spec: MLPTrialSpec, device: torch.device, seed: int, trial_index: int, low_precision: bool):
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
        self.Gf = torch.zeros(off, device=device, dtype=wdtype)      # flat gradients
        self.params: List[torch.Tensor] = []
        for w_off, b_off, o, i in self.offsets:
            w = self.W[w_off:w_off + o * i].view(o, i).requires_grad_(True)
            b = self.W[b_off:b_off + o].requires_grad_(True)
            w.grad = self.Gf[w_off:w_off + o * i].view(o, i)
            b.grad = self.Gf[b_off:b_off + o]
            self.params += [w, b]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
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
        if comm.world_size > 1:
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
                 comm: Optional[Comm] = None, seed: int = 0, trial_index: int = 0):
        self.comm = comm or LocalComm(X.device)
        self.device = X.device
        self.spec = MLPTrialSpec.from_config(config, int(X.shape[1]))
        self.gpu = self.device.type == "cuda"
        self.X, self.y, self.Xval, self.yval = X, y, Xval, yval
        if self.gpu:        # the table is kept in bf16: half the HBM, and it is what the GEMMs consume
            self.X = X if X.dtype == torch.bfloat16 else X.to(torch.bfloat16)
            self.Xval = Xval if Xval.dtype == torch.bfloat16 else Xval.to(torch.bfloat16)
        self.model = FlatMLP(self.spec, self.device, seed, trial_index, low_precision=self.gpu)
        self.reducer = BucketedAllReduce(self.model, self.comm)
        self.seed, self.trial_index = seed, trial_index
        self.global_batch = self.spec.batch_size
        if self.global_batch % self.comm.world_size:
            raise ValueError("batch_size must be divisible by the number of ranks")
        self.local_batch = self.global_batch // self.comm.world_size
        self.epoch, self.step = 0, 0
        self.stats = {"steps": 0, "train_s": 0.0}
        if self.gpu:
            from ..ops import lib as L
            self.L, self.lib = L, L.load()
            self.M = torch.zeros_like(self.model.P)
            self.V = torch.zeros_like(self.model.P)
            self.ids = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._hp_t = None
        else:
            self.opt = build_optimizer(self.spec.optimizer, [self.model.P], self.spec.learning_rate,
                                       self.spec.weight_decay, self.spec.momentum, self.spec.beta1, self.spec.beta2,
                                       self.spec.eps)

    # ------------------------------------------------------------------ optimiser
    def _apply_update(self) -> None:
        sp = self.spec
        if self.gpu:
            if self._hp_t is None:
                hp = np.zeros(1, dtype=self.L.OPT_DTYPE)
                hp[0] = (sp.learning_rate, sp.beta1, sp.beta2, sp.eps, sp.weight_decay, sp.momentum, 0.0,
                         OPT_CODES[sp.optimizer], sp.warmup_steps, sp.total_steps, self.model.numel, 1 << 30, 0,
                         0, 0, 0)
                self._hp_t = torch.from_numpy(hp.view(np.uint8).reshape(-1).copy()).to(self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.L.check(self.lib.mt_optim_launch_bf16g(
                self.model.P.data_ptr(), self.model.Gf.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                self.model.W.data_ptr(), _round_up(self.model.numel, 8), self._hp_t.data_ptr(), self.ids.data_ptr(), 1,
                self.model.numel, self.step, stream), "mt_optim_launch_bf16g")
        else:
            for g in self.opt.param_groups:
                g["lr"] = sp.learning_rate * lr_factor(self.step, sp.warmup_steps, sp.total_steps)
            self.model.P.grad = self.model.Gf
            self.opt.step()

    # ------------------------------------------------------------------ epoch
    def train_epoch(self) -> float:
        t0 = time.time()
        n_local = int(self.X.shape[0])
        steps = n_local // self.local_batch
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
    trainer = DPTrainer(config, train[0], train[1], val[0], val[1], comm=comm, seed=seed,
                        trial_index=int(tid) if tid.isdigit() else 0)
    for _ in range(int(config.get("epochs", epochs))):
        train_loss = trainer.train_epoch()
        metrics = trainer.evaluate()
        metrics["train_loss"] = train_loss
        metrics["allreduce_gb"] = trainer.reducer.bytes_reduced / 1e9
        session.report(metrics)
