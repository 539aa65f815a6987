"""Single-trial training on the GPU without per-step host work: flat parameters, a device-state optimiser and a
hipGraph-captured training step.

The reference's loop (/root/reference/ray-tune-hpo-regression-sample.py:110-120 and
/root/reference/ray-tune-hpo-regression.py:322-345) pays, per optimiser step, a DataLoader batch on the CPU, a
host-to-device copy, a few hundred eager kernel launches of a tiny model and a ``loss.item()`` synchronisation.
For models this small the GPU idles between launches.  Here:

* the dataset lives on the device; a batch is an ``index_select`` with indices taken from a device-resident
  permutation at a device-resident cursor;
* all parameters are views of ONE f32 slab and all gradients views of a twin slab (``FlatParams``), so the
  optimiser is one launch of ``csrc/flat_optim.hip`` whose learning rate, schedule and step counter are read
  from device memory (``FlatOptimizer``) -- nothing about a step is a host-side value;
* forward + loss + backward + clip + update + cursor advance are captured ONCE into a hipGraph and replayed for
  every batch of every epoch (``GraphedTrainer``); the host reads one scalar per epoch.

Dropout keeps drawing fresh masks under replay because every random op inside the step takes its seed from a
device-resident counter that the captured step itself advances (``ops/rng.py``); each trainer owns its counter, so
several trainers may replay their graphs on different streams at once.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import gc
import os
import logging
import threading
from typing import Callable, Dict, Optional

import torch

from .optim import OPT_CODES
from ..ops import rng
from ..options import OPTIONS

log = logging.getLogger("mitune")

# one capture at a time per process
_CAPTURE_LOCK = threading.Lock()
# Set by the thread executor when several trial threads share one GPU: HIP stream capture is invalidated by
# device work issued from sibling threads, so such trials step eagerly (each on its own stream) instead.
SHARED_GPU_THREADS = False

# Signatures whose kernels have all run once on the calling thread, and ONE stream per device on which every warm-up and
# capture is issued: the BLAS library keeps a workspace per (handle, stream), and on a stream that already has one a
# capture never allocates it from a graph's private pool.
_WARM_LOCAL = threading.local()        # thread-local, not keyed by thread id: the OS recycles ids of finished threads


def _warmed_on_this_thread() -> set:
    w = getattr(_WARM_LOCAL, "signatures", None)
    if w is None:
        w = _WARM_LOCAL.signatures = set()
    return w

_CAPTURE_STREAMS: Dict = {}


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing: a collection may destroy the graph of a trainer that
    died earlier, and destroying a graph (or releasing its memory pool) inside a capture aborts the process.  (The
    framework's own capture context collects *before* every capture instead, which costs milliseconds per trial.)"""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _capture_stream(device) -> "torch.cuda.Stream":
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _CAPTURE_STREAMS.get(key)
    if st is None:
        st = _CAPTURE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


_ALIGN = 8        # parameter offsets in the slab are multiples of 8 floats (32 bytes)
_SEG = 2048       # elements per optimiser block / gradient-table segment


class FlatParams:
    """Re-homes a module's trainable parameters into one contiguous f32 slab.

    ``grad_slab=True`` also binds every gradient to a view of a twin slab (autograd then accumulates in place, one add
    kernel per parameter); ``grad_slab=False`` leaves gradients unbound, so autograd hands over the tensors its backward
    ops produced and the optimiser reads them where they are (``FlatOptimizer`` keeps a table of their addresses).
    """

    def __init__(self, model: torch.nn.Module, grad_slab: bool = True, bf16_shadow: bool = False, device=None):
        """``device``: where the slab lives.  A model still on the CPU is moved there by ONE upload of the flattened
        parameters (a module's own ``.to(device)`` issues a blocking copy per parameter); buffers follow one by one.

        ``bf16_shadow``: also keep a bf16 copy of the slab (every parameter gets a ``_mt_bf16`` view of it), refreshed
        by ``refresh_shadow()`` after each update -- for bf16-autocast models whose dense layers would otherwise cast
        their weight and bias on every call."""
        self.grad_slab = grad_slab
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("model has no trainable parameters")
        src = self.params[0].device
        dev = torch.device(device) if device is not None else src
        offs, n = [], 0
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("FlatParams needs f32 parameters")
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n, self.offsets = n, offs
        self.G = torch.zeros(n if grad_slab else 0, device=dev)
        with torch.no_grad():
            if src.type == "cpu" and dev.type != "cpu":
                host = torch.zeros(n)
                for p, o in zip(self.params, offs):
                    host[o:o + p.numel()] = p.detach().reshape(-1)
                self.P = host.to(dev)                        # one upload
                frozen = [q for q in model.parameters() if not q.requires_grad]
                for t in frozen:
                    t.data = t.data.to(dev)
                for mod in model.modules():                  # buffers (e.g. the positional-encoding table)
                    for name, buf in list(mod._buffers.items()):
                        if buf is not None:
                            mod._buffers[name] = buf.to(dev)
            else:
                self.P = torch.zeros(n, device=dev)
                for p, o in zip(self.params, offs):
                    self.P[o:o + p.numel()].copy_(p.detach().reshape(-1))
            for p, o in zip(self.params, offs):
                p.data = self.P[o:o + p.numel()].view(p.shape)
                # slab mode: autograd accumulates in place into the slab
                p.grad = self.G[o:o + p.numel()].view(p.shape) if grad_slab else None
        self.Pb: Optional[torch.Tensor] = None
        if bf16_shadow:
            self.Pb = self.P.to(torch.bfloat16)
            for p, o in zip(self.params, offs):
                p._mt_bf16 = self.Pb[o:o + p.numel()].view(p.shape)

    def refresh_shadow(self) -> None:
        if self.Pb is not None:
            self.Pb.copy_(self.P)


class FlatOptimizer:
    """Adam / AdamW / SGD(momentum) / RMSprop(momentum) over a ``FlatParams`` slab, state on the device."""

    def __init__(self, flat: FlatParams, name: str = "adamw", lr: float = 1e-3, weight_decay: float = 0.0,
                 momentum: float = 0.9, betas=(0.9, 0.999), eps: float = 1e-8, clip: float = 0.0, warmup: int = 0,
                 total: int = 0):
        from ..ops import lib as L
        self.so = L.load()
        self.L = L
        self.flat = flat
        self.name = name.lower()
        dev = flat.P.device
        self.M = torch.zeros_like(flat.P)
        self.V = torch.zeros_like(flat.P)
        rec = L.FlatOpt(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps, wd=weight_decay,
                        momentum=momentum if self.name in ("sgd", "rmsprop") else 0.0, clip=max(0.0, clip),
                        opt_type=OPT_CODES[self.name], warmup=int(warmup), total=int(total), step=0, pad=0)
        self.state = torch.frombuffer(bytearray(bytes(rec)), dtype=torch.uint8).to(dev)
        self._f32 = self.state.view(torch.float32)
        self._i32 = self.state.view(torch.int32)
        self.use_clip = clip > 0.0
        nb = int(self.so.mt_flat_optim_blocks(flat.n))
        self._segs = None
        if not flat.grad_slab:
            # one segment per block of up to 2048 elements of a parameter: [slab offset, gradient address, count]
            rows, owner, local = [], [], []
            for i, (p, off) in enumerate(zip(flat.params, flat.offsets)):
                for c in range(0, p.numel(), _SEG):
                    rows.append([off + c, 0, min(_SEG, p.numel() - c)])
                    owner.append(i)
                    local.append(c * 4)
            self._segs = torch.tensor(rows, dtype=torch.int64)
            self._seg_owner = torch.tensor(owner, dtype=torch.int64)
            self._seg_local = torch.tensor(local, dtype=torch.int64)
            self._tables = {"eager": self._segs.to(dev), "graph": self._segs.to(dev)}
            self._ptrs = {"eager": None, "graph": None}
            self._deferred = None
            nb = len(rows)
        self.partials = torch.zeros(max(nb, 1), device=dev)

    # field slots of the FlatOpt record (4-byte words)
    _LR, _WD, _STEP = 0, 4, 10

    def set_hyper(self, lr: Optional[float] = None, weight_decay: Optional[float] = None) -> None:
        if lr is not None:
            self._f32[self._LR:self._LR + 1].fill_(float(lr))
        if weight_decay is not None:
            self._f32[self._WD:self._WD + 1].fill_(float(weight_decay))

    def _write_table(self, which: str, ptrs) -> None:
        base = torch.tensor(ptrs, dtype=torch.int64)[self._seg_owner]
        host = self._segs.clone()
        host[:, 1] = torch.where(base != 0, base + self._seg_local, base)
        self._tables[which].copy_(host)
        self._ptrs[which] = ptrs

    def finish_capture(self) -> None:
        """Fill the captured step's gradient table (its addresses are only known once the capture has run)."""
        if self._segs is not None and self._deferred is not None:
            self._write_table("graph", self._deferred)
            self._deferred = None

    def _step_scattered(self) -> None:
        f = self.flat
        ptrs = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in f.params)
        stream = torch.cuda.current_stream(f.P.device)
        if torch.cuda.is_current_stream_capturing():
            which, self._deferred = "graph", ptrs
        else:
            which = "eager"
            if ptrs != self._ptrs["eager"]:
                self._write_table("eager", ptrs)
        table = self._tables[which]
        self.L.check(self.so.mt_flat_optim_seg_launch(f.P.data_ptr(), self.M.data_ptr(), self.V.data_ptr(), table.data_ptr(),
                                                      table.shape[0], self.state.data_ptr(), self.partials.data_ptr(),
                                                      1 if self.use_clip else 0, stream.cuda_stream),
                     "mt_flat_optim_seg_launch")
        for p in f.params:
            p.grad = None            # the next backward hands over fresh tensors (same addresses under graph replay)
        f.refresh_shadow()

    def step(self) -> None:
        f = self.flat
        if self._segs is not None:
            return self._step_scattered()
        self.L.check(self.so.mt_flat_optim_launch(f.P.data_ptr(), f.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr(),
                                                  f.n, self.state.data_ptr(), self.partials.data_ptr(),
                                                  1 if self.use_clip else 0, 1,
                                                  torch.cuda.current_stream(f.P.device).cuda_stream),
                     "mt_flat_optim_launch")
        f.refresh_shadow()

    def zero_grad(self) -> None:      # the update kernel already zeroes the slab; this is for completeness
        if self._segs is not None:
            for p in self.flat.params:
                p.grad = None
        else:
            self.flat.G.zero_()

    @property
    def steps_taken(self) -> int:
        return int(self._i32[self._STEP].item())

    def state_dict(self) -> Dict:
        return {"flat": True, "name": self.name, "record": self.state.cpu().clone(), "M": self.M.cpu().clone(),
                "V": self.V.cpu().clone()}

    def load_state_dict(self, sd: Dict) -> None:
        if not sd.get("flat") or sd["M"].numel() != self.M.numel() or sd.get("name") != self.name:
            raise ValueError("optimizer state does not match this flat optimizer")
        self.M.copy_(sd["M"])
        self.V.copy_(sd["V"])
        self._i32[self._STEP:self._STEP + 1].copy_(sd["record"].view(torch.int32)[self._STEP:self._STEP + 1])


class GraphedTrainer:
    """Epochs of graph-replayed optimiser steps over a device-resident dataset.

    ``X`` [N, ...] and ``Y`` [N, ...] must already be on the model's device.  One call of ``train_epoch`` shuffles
    (a device ``randperm``), replays the captured step ``N // batch_size`` times, runs a ragged last batch eagerly
    (kept, like a DataLoader without ``drop_last``) and returns the mean batch loss with one synchronisation.
    """

    def __init__(self, model: torch.nn.Module, criterion: Callable, X: torch.Tensor, Y: torch.Tensor, batch_size: int,
                 optimizer: FlatOptimizer, shuffle: bool = True, seed: int = 0, use_graph: bool = True,
                 drop_last: bool = False, forward_ctx: Optional[Callable] = None, signature=None):
        """``signature``: a hashable description of everything that decides WHICH kernels a step launches (model family,
        parameter shapes, batch, optimiser, loss, which regularisers are active).  The first trainer of a signature runs
        one eager step before capturing (lazy initialisation must not happen inside a capture); later ones capture
        straight away.  None: always warm up."""
        self.signature = signature
        self.model, self.criterion, self.opt = model, criterion, optimizer
        self.drop_last = drop_last
        self.forward_ctx = forward_ctx or contextlib.nullcontext      # e.g. a bf16 autocast factory
        self.X, self.Y = X, Y
        self.N, self.B = int(X.shape[0]), int(min(batch_size, X.shape[0]))
        dev = X.device
        self.shuffle = shuffle
        self.gen = torch.Generator(device=dev).manual_seed(int(seed))
        self.perm = torch.arange(self.N, device=dev)
        self.ar = torch.arange(self.B, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)
        self.running = torch.zeros((), device=dev)
        self.rng = rng.new_counter(dev, 0x1234567 + int(seed)) if dev.type == "cuda" else None
        self._fast = False
        if dev.type == "cuda":
            from ..ops.batch_hip import cursor_batch_ok
            self._fast = cursor_batch_ok(self.perm, self.pos, X, Y) and OPTIONS.fast_batch
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graphed = False
        if use_graph and not SHARED_GPU_THREADS and dev.type == "cuda" and self.N >= self.B:
            try:
                self._capture()
                self.graphed = True
            except Exception as e:       # noqa: BLE001 - fall back to eager stepping, loudly
                log.warning("hipGraph capture of the training step failed (%s: %s); stepping eagerly",
                            type(e).__name__, e)
                self.graph = None
                torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ one optimiser step
    def _step(self, rows: torch.Tensor) -> None:
        self._step_xy(self.X.index_select(0, rows), self.Y.index_select(0, rows))

    def _step_xy(self, x: torch.Tensor, y: torch.Tensor, advance: bool = False) -> bool:
        """One optimiser step on the batch (x, y).  ``advance``: the epoch cursor should move on by one batch; returns
        True when that was folded into the fused loss launch (else the caller advances it)."""
        fused_mse = self._fast and type(self.criterion) is torch.nn.MSELoss and self.criterion.reduction == "mean"
        with rng.scope(self.rng):
            rng.begin_step(x.device)          # the seed table is refilled inside the (captured) step
            with self.forward_ctx():
                out = self.model(x)
            out = out.float().reshape(y.shape)
            if fused_mse:                     # loss, its gradient and the running sum in one launch
                from ..ops.batch_hip import FusedMSE
                loss = FusedMSE.apply(out, y, self.running, self.pos if advance else None, self.B)
            else:
                loss = self.criterion(out, y)
            loss.backward()
        self.opt.step()
        if not fused_mse:
            self.running += loss.detach()
        return fused_mse and advance

    def seed(self, value: int) -> None:
        """Pin this trainer's dropout counter (stream-ordered)."""
        if self.rng is not None:
            self.rng.seed(value)

    def _cursor_step(self) -> None:
        if self._fast:                        # batch rows picked at the device cursor in one launch
            from ..ops.batch_hip import cursor_batch
            if self._step_xy(*cursor_batch(self.perm, self.pos, self.X, self.Y, self.B), advance=True):
                return
        else:
            self._step(self.perm.index_select(0, self.ar + self.pos))
        self.pos += self.B

    def _capture(self) -> None:
        f, o = self.opt.flat, self.opt
        cur = torch.cuda.current_stream()
        side = _capture_stream(self.X.device)
        self.model.train()
        # per thread: BLAS handles are per thread, and a handle's first use must not fall inside a capture
        warmed = _warmed_on_this_thread()
        if self.signature is None or self.signature not in warmed:
            keep = (f.P.clone(), o.M.clone(), o.V.clone(), o.state.clone(),
                    None if self.rng is None else self.rng.ctr.clone())
            side.wait_stream(cur)
            with torch.cuda.stream(side):      # warm-up off the caller's stream (allocator, lazy init, library workspaces)
                self.pos.zero_()
                self._cursor_step()
            cur.wait_stream(side)
            with torch.no_grad():              # undo the warm-up step: training starts from the initial weights
                f.P.copy_(keep[0]); o.M.copy_(keep[1]); o.V.copy_(keep[2]); o.state.copy_(keep[3])
                f.refresh_shadow()
                o.zero_grad(); self.pos.zero_(); self.running.zero_()
                if self.rng is not None:
                    self.rng.ctr.copy_(keep[4])
        g = torch.cuda.CUDAGraph()
        # Raw capture_begin / capture_end on the warm-up stream: the torch.cuda.graph context would first synchronise the
        # device, run the Python garbage collector and empty the allocator cache -- tens of milliseconds per trial that a
        # search over hundreds of sub-second trials pays again and again (and every trial after it re-mallocs).
        with _CAPTURE_LOCK, _no_gc():
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    self._cursor_step()
                finally:
                    g.capture_end()
            cur.wait_stream(side)
        o.finish_capture()
        self.graph = g
        if self.signature is not None:
            warmed.add(self.signature)

    def capture_callable(self, fn: Callable[[], None], warm: bool = False) -> Optional["torch.cuda.CUDAGraph"]:
        """Capture ``fn()`` (device work only, fixed addresses) into a hipGraph on a side stream; None if graphs are off
        or the capture fails.  ``warm`` runs it once eagerly first (first use of its kernels in the process)."""
        if self.graph is None:
            return None
        cur = torch.cuda.current_stream()
        side = _capture_stream(self.X.device)
        try:
            if warm:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    fn()
                cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with _CAPTURE_LOCK, _no_gc():
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    g.capture_begin(capture_error_mode="thread_local")
                    try:
                        fn()
                    finally:
                        g.capture_end()
                cur.wait_stream(side)
            return g
        except Exception as e:       # noqa: BLE001 - the caller keeps its eager path
            log.warning("hipGraph capture of an auxiliary pass failed (%s: %s); it stays eager", type(e).__name__, e)
            torch.cuda.synchronize(self.X.device)
            return None

    # ------------------------------------------------------------------ epochs
    def begin_epoch(self) -> int:
        """Shuffle, rewind the cursor; returns how many ``step_once`` calls make up the epoch."""
        self.model.train()
        if self.shuffle:
            self.perm.copy_(torch.randperm(self.N, device=self.perm.device, generator=self.gen))
        self.pos.zero_()
        self.running.zero_()
        n_full = self.N // self.B
        return n_full + (1 if self.N % self.B and not self.drop_last else 0)

    def step_once(self, i: int) -> None:
        """Step ``i`` of the epoch on the current stream: a graph replay, or the eager ragged last batch."""
        n_full = self.N // self.B
        if i >= n_full:
            self._step(self.perm[n_full * self.B:])
        elif self.graph is not None:
            self.graph.replay()
        else:
            self._cursor_step()

    def epoch_loss(self, batches: int) -> torch.Tensor:
        """Mean batch loss of the epoch as a device scalar (no synchronisation)."""
        return self.running / max(batches, 1)

    def train_epoch(self) -> float:
        batches = self.begin_epoch()
        for i in range(batches):
            self.step_once(i)
        return float(self.epoch_loss(batches))

    @torch.no_grad()
    def _validation_sums(self, Xv: torch.Tensor, Yv: torch.Tensor, chunk: int, mape_eps: float, out2: torch.Tensor) -> None:
        """[sum of per-sample losses, sum of absolute percentage errors] over the validation set -> ``out2`` (f64)."""
        was_training = self.model.training
        self.model.eval()
        n = int(Xv.shape[0])
        out2.zero_()
        for c0 in range(0, n, chunk):
            x, y = Xv[c0:c0 + chunk], Yv[c0:c0 + chunk]
            with self.forward_ctx():
                out = self.model(x)
            out = out.float().reshape(y.shape)
            out2 += torch.stack([self.criterion(out, y).double() * y.numel(),
                                 (torch.abs(y - out) / (torch.abs(y) + mape_eps)).sum().double()])
        self.model.train(was_training)

    def evaluate(self, Xv: torch.Tensor, Yv: torch.Tensor, chunk: int = 4096, mape_eps: float = 1e-8) -> Dict[str, float]:
        """Sample-weighted validation loss and MAPE over the whole validation set in large chunks."""
        n = int(Xv.shape[0])
        acc = torch.zeros(2, device=Xv.device, dtype=torch.float64)
        self._validation_sums(Xv, Yv, chunk, mape_eps, acc)
        both = acc.tolist()                                    # one synchronisation
        return {"validation_loss": both[0] / max(n, 1), "validation_mape": 100.0 * both[1] / max(n, 1)}

    def epoch_with_validation(self, Xv: torch.Tensor, Yv: torch.Tensor, chunk: int = 4096,
                              mape_eps: float = 1e-8) -> Dict[str, float]:
        """One training epoch followed by the validation pass, read back with ONE synchronisation.  With graphs on, the
        validation pass over a small set is captured at the first call and replayed behind the last training step."""
        n = int(Xv.shape[0])
        key = (Xv.data_ptr(), Yv.data_ptr(), n, chunk)
        if getattr(self, "_val_key", None) != key:
            self._val_key, self._val_acc = key, torch.zeros(2, device=Xv.device, dtype=torch.float64)
            self._val_graph = None
            if self.graph is not None and min(n, chunk) * max(1, Xv[0].numel() // max(1, Xv.shape[-1])) <= 16384:
                self._val_graph = self.capture_callable(
                    lambda: self._validation_sums(Xv, Yv, chunk, mape_eps, self._val_acc), warm=True)
        batches = self.begin_epoch()
        for i in range(batches):
            self.step_once(i)
        if self._val_graph is not None:
            self._val_graph.replay()
        else:
            self._validation_sums(Xv, Yv, chunk, mape_eps, self._val_acc)
        vals = torch.cat([self.epoch_loss(batches).double().reshape(1), self._val_acc]).tolist()
        return {"train_loss": vals[0], "validation_loss": vals[1] / max(n, 1), "validation_mape": 100.0 * vals[2] / max(n, 1)}
