"""SeqPackEngine: a pack of small Transformer-encoder trials trained in shared launches.

The reference's sample experiment (/root/reference/ray-tune-hpo-regression-sample.py:60-135) trains one
``SimpleTransformerModel`` per trial, one trial per GPU: batch 32 x 50 tokens, d_model 32-128 -- roughly 55 kernels
of a few workgroups each per step, on a chip with 256 compute units.  ``GraphPackEngine`` shares the GPU between such
trials through HIP streams (one captured step per trial); this engine shares the LAUNCHES: all resident trials,
whatever their width / heads / depth / dropout / learning rate, advance by one step in 14 launches per encoder layer
plus 6, each launch covering every trial (``ops/csrc/seq.hip`` has the launch list).

Memory model: one f32 parameter slab ``P`` (+ ``M``, ``V`` and ``ksplits`` gradient slabs ``G``) with a fixed stride
per slot, laid out per trial by ``SeqLayout``; one activation arena with a fixed stride per slot holding the slot's
mini-batch, every tensor the backward pass needs (projection outputs, pre-norm sums, row statistics) and the gradient
buffers that are reused layer after layer.  Nothing is allocated after construction.

Contract shared with the other engines (checked in tests against ``TorchPackEngine``): initial weights of a trial come
from the trial's own seeded stream, batches from the keyed permutation of (seed, trial, epoch), an epoch is
``N // batch`` steps, validation metrics are sample-weighted sums over the whole validation set.  Arithmetic is f32
throughout: exact-f32 MFMA for the dense layers, f32 multiply-add chains in the attention kernels for sequences of up to
128 tokens (the same kernels one-trial-at-a-time training uses; ``OPTIONS.attn_exact`` 0 or 1 selects the bf16
matrix form for all, or for the longer, sequences instead).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import PackEngine
from .engine_hip import SMALL_TILE, balanced_tiles
from .permutation import half_bits, mix32, round_keys
from .trainable import PackedTrainable
from ..models.mlp import MLPTrialSpec
from ..ops import lib as L
from ..options import OPTIONS
from ..train.losses import LOSS_CODES
from ..train.optim import OPT_CODES

MAX_LAYERS = 8
# host threads drawing the next trials' weights while the GPU trains (1 / 2 / 4 threads on the 2048-trial search: 6.27 /
# 6.19 / 6.64 s -- more threads fight over the interpreter lock between the many small fills)
# (OPTIONS.seq_prep_threads; OPTIONS.seq_ksplits: weight-gradient slabs, the token contraction is split in this many)


def _chain_rule(text: str):
    return sorted((tuple(int(v) for v in part.split(":")) for part in text.split(",") if part), reverse=True)


# "min_slots:chains,...": packs of at least min_slots trials are split into that many independent chains of launches on
# separate HIP streams (the first rule reached applies; none: one chain).  Measured per pack step, one chain / two
# chains / four chains: 16 slots 907 / 985 / 970 us, 64 slots 2236 / 2100 / 2202 us, 128 slots 3965 / 3702 / 4131 us --
# two chains fill each other's launch tails once a pack is big enough to keep both busy.
# (OPTIONS.seq_chain_rule)


def ceil4(x: int) -> int:
    return (x + 3) // 4 * 4


def ceil8(x: int) -> int:
    return (x + 7) // 8 * 8


c_void_p, c_int32 = C.c_void_p, C.c_int32


class SeqOp(C.Structure):
    _fields_ = [("groups", c_void_p), ("tiles", c_void_p), ("ntiles", c_int32), ("edge", c_int32)]


class SeqPlan(C.Structure):
    """Mirror of ``MtSeqPlan`` (csrc/seq.hip)."""
    _fields_ = [("n_layers", c_int32), ("n_active", c_int32), ("S", c_int32), ("max_b", c_int32), ("max_bh", c_int32),
                ("max_m", c_int32), ("max_params", c_int32), ("wide", c_int32),
                ("row_len", c_int32), ("hb", c_int32), ("use_clip", c_int32), ("ksplits", c_int32),
                ("max_blocks", c_int32), ("pad0", c_int32), ("n_rows", C.c_uint32), ("pad1", C.c_uint32),
                ("X", c_void_p), ("Y", c_void_p), ("gather", c_void_p), ("ids", c_void_p),
                ("in_proj", SeqOp),
                ("qkv", SeqOp * MAX_LAYERS), ("outp", SeqOp * MAX_LAYERS), ("ff1", SeqOp * MAX_LAYERS),
                ("ff2", SeqOp * MAX_LAYERS),
                ("attn", c_void_p * MAX_LAYERS), ("ln1", c_void_p * MAX_LAYERS), ("ln2", c_void_p * MAX_LAYERS),
                ("d_ff2", SeqOp * MAX_LAYERS), ("d_ff1", SeqOp * MAX_LAYERS), ("d_out", SeqOp * MAX_LAYERS),
                ("d_qkv", SeqOp * MAX_LAYERS),
                ("wgrad", SeqOp), ("head", c_void_p),
                ("P", c_void_p), ("G", c_void_p), ("M", c_void_p), ("V", c_void_p),
                ("slot_stride", C.c_size_t), ("gsplit_stride", C.c_size_t), ("opt", c_void_p), ("partials", c_void_p)]


ATTN_DTYPE = np.dtype([("q", "<u8"), ("k", "<u8"), ("v", "<u8"), ("o", "<u8"), ("stat", "<u8"), ("d_o", "<u8"),
                       ("dq", "<u8"), ("dk_", "<u8"), ("dv_", "<u8"), ("seed", "<u8"),
                       ("B", "<i4"), ("H", "<i4"), ("dk", "<i4"), ("dv", "<i4"),
                       ("ldq", "<i4"), ("ldv", "<i4"), ("ldo", "<i4"), ("kind", "<i4"),
                       ("active_steps", "<i4"), ("site", "<i4"),
                       ("scale", "<f4"), ("keep_scale", "<f4"), ("thresh", "<u4"), ("pad", "<i4")], align=True)


def attn_slot(q: int, k: int, v: int, o: int, stat: int, d_o: int, dq: int, dk_: int, dv_: int, seed: int, B: int, H: int,
              dk: int, dv: int, ldq: int, ldv: int, ldo: int, kind: int, steps: int, site: int, scale: float, keep: float,
              thresh: int) -> tuple:
    """One ``MtAttnSlot`` record (csrc/attention.hip)."""
    return (q, k, v, o, stat, d_o, dq, dk_, dv_, seed, B, H, dk, dv, ldq, ldv, ldo, kind, steps, site, scale, keep, thresh, 0)
LN_DTYPE = np.dtype([("x", "<u8"), ("res", "<u8"), ("w", "<u8"), ("b", "<u8"), ("y", "<u8"), ("z", "<u8"),
                     ("stats", "<u8"), ("dy", "<u8"), ("dx", "<u8"), ("dres", "<u8"), ("part", "<u8"), ("gw", "<u8"),
                     ("gb", "<u8"), ("seed", "<u8"),
                     ("M", "<i4"), ("d", "<i4"), ("active_steps", "<i4"), ("site", "<i4"),
                     ("keep_scale", "<f4"), ("eps", "<f4"), ("thresh", "<u4"), ("S", "<i4"),
                     ("sd_thresh", "<u4"), ("sd_scale", "<f4"), ("sd_site", "<i4"), ("pad", "<i4")], align=True)
SEQ_HEAD_DTYPE = np.dtype([("H", "<u8"), ("dH", "<u8"), ("w", "<u8"), ("b", "<u8"), ("gw", "<u8"), ("gb", "<u8"),
                           ("y", "<u8"), ("loss_acc", "<u8"), ("metrics", "<u8"),
                           ("B", "<i4"), ("d", "<i4"), ("loss_type", "<i4"), ("active_steps", "<i4")], align=True)

_bound = False


def _bind(lib) -> None:
    """Type the sequence-pack entry points and check the struct layouts against the library (once)."""
    global _bound
    if _bound:
        return
    sizes = [(lib.mt_seq_plan_sizeof(), C.sizeof(SeqPlan), "MtSeqPlan"),
             (lib.mt_seq_ln_slot_sizeof(), LN_DTYPE.itemsize, "MtLnSlot"),
             (lib.mt_seq_head_slot_sizeof(), SEQ_HEAD_DTYPE.itemsize, "MtSeqHeadSlot"),
             (lib.mt_attn_slot_sizeof(), ATTN_DTYPE.itemsize, "MtAttnSlot")]
    for got, want, name in sizes:
        if got != want:
            raise L.KernelLibraryError(f"{name} is {got} bytes in the library, {want} in engine_seq.py")
    lib.mt_seq_train_steps.argtypes = [C.POINTER(SeqPlan), C.c_int, C.c_int, c_void_p]
    lib.mt_seq_eval.argtypes = [C.POINTER(SeqPlan), C.c_int, C.c_int, c_void_p]
    lib.mt_seq_train_steps.restype = lib.mt_seq_eval.restype = C.c_int
    _bound = True


@dataclasses.dataclass
class SeqSpec:
    d: int
    heads: int
    layers: int
    dropout: float
    opt: MLPTrialSpec          # batch size, optimiser family and its numbers, loss, schedule

    @property
    def dk(self) -> int:
        return self.d // self.heads


class SeqLayout:
    """Element offsets of one trial's parameter tensors inside its slab region (every tensor starts on a multiple of 8
    floats; the input projection's rows are padded to ``fpad`` columns, pad columns are zero and stay zero)."""

    def __init__(self, F: int, fpad: int, d: int, layers: int):
        self.F, self.fpad, self.d, self.layers = F, fpad, d, layers
        self.entries: List[Tuple[str, int, Tuple[int, int], int]] = []      # key, offset, (rows, cols), row stride
        self.off: Dict[str, int] = {}
        at = 0

        def add(key: str, rows: int, cols: int, ld: Optional[int] = None):
            nonlocal at
            ld = ld or cols
            self.entries.append((key, at, (rows, cols), ld))
            self.off[key] = at
            at = ceil8(at + rows * ld)

        add("input_projection.weight", d, F, fpad)
        add("input_projection.bias", 1, d)
        for l in range(layers):
            p = f"transformer.{l}."
            add(p + "self_attn.qkv_proj.weight", 3 * d, d)
            add(p + "self_attn.qkv_proj.bias", 1, 3 * d)
            add(p + "self_attn.out_proj.weight", d, d)
            add(p + "self_attn.out_proj.bias", 1, d)
            add(p + "linear1.weight", 4 * d, d)
            add(p + "linear1.bias", 1, 4 * d)
            add(p + "linear2.weight", d, 4 * d)
            add(p + "linear2.bias", 1, d)
            add(p + "norm1.weight", 1, d)
            add(p + "norm1.bias", 1, d)
            add(p + "norm2.weight", 1, d)
            add(p + "norm2.bias", 1, d)
        add("fc.weight", 1, d)
        add("fc.bias", 1, 1)
        self.total = at


class ArenaLayout:
    """Float offsets of one slot's activation / gradient buffers for ``rows`` samples of ``S`` tokens."""

    def __init__(self, rows: int, S: int, fpad: int, d: int, heads: int, layers: int):
        m = rows * S
        self.off: Dict[str, int] = {}
        at = 0

        def add(key: str, n: int):
            nonlocal at
            self.off[key] = at
            at = ceil4(at + n)

        add("X0", m * fpad)
        add("y", rows)
        add("H0", m * d)
        add("T", m * d)
        for l in range(layers):
            add(f"QKV{l}", m * 3 * d)
            add(f"O{l}", m * d)
            add(f"Z1_{l}", m * d)
            add(f"H1_{l}", m * d)
            add(f"F1_{l}", m * 4 * d)
            add(f"Z2_{l}", m * d)
            add(f"H2_{l}", m * d)
            add(f"st1_{l}", 2 * m)
            add(f"st2_{l}", 2 * m)
            add(f"astat{l}", rows * heads * S)
            add(f"part1_{l}", ((m + 31) // 32) * 2 * d)
            add(f"part2_{l}", ((m + 31) // 32) * 2 * d)
        # gradients the weight-gradient launch at the end of the backward pass still reads: one buffer per layer
        for l in range(layers):
            add(f"dB2_{l}", m * d)          # gradient of the feed-forward output (after its dropout mask)
            add(f"dF1_{l}", m * 4 * d)
            add(f"dB1_{l}", m * d)          # gradient of the attention projection's output
            add(f"dQKV{l}", m * 3 * d)
        for key, width in (("dH", d), ("dR", d), ("dO", d)):      # consumed inside the chain: reused layer after layer
            add(key, m * width)
        self.total = at


class _SSlot:
    __slots__ = ("trial_index", "spec", "layout", "arena", "epoch", "step", "config")


def seq_spec(trainable: PackedTrainable, config: Dict, n_train: int) -> SeqSpec:
    merged = {**trainable.model_kwargs, **config}
    opt = MLPTrialSpec.from_config(merged, trainable.in_features)
    if opt.warmup_steps or opt.total_steps:      # same rule as the other engines: the decay never ends before training does
        planned = trainable.max_epochs * max(1, n_train // max(1, min(opt.batch_size, n_train)))
        opt = dataclasses.replace(opt, total_steps=max(opt.total_steps, planned) if opt.total_steps else 0,
                                  warmup_steps=min(opt.warmup_steps, max(1, planned // 4)) if opt.warmup_steps else 0)
    return SeqSpec(int(merged["d_model"]), int(merged["num_heads"]), int(merged["num_layers"]),
                   float(merged.get("dropout", 0.1)), opt)


def supports(trainable: PackedTrainable, config: Optional[Dict] = None) -> Optional[str]:
    """None if this engine can train the trainable (and, when given, the trial config); else the reason it cannot."""
    if trainable.model != "simple_transformer":
        return "the sequence-pack engine trains the 'simple_transformer' family"
    X = trainable.train[0]
    if X.dim() != 3:
        return "sequence data [N, S, F] is required"
    if X.shape[1] > 128:
        return "sequences longer than 128 tokens do not fit the one-workgroup attention kernel"
    if trainable.dtype not in ("f32", "bf16"):
        return f"unknown dtype {trainable.dtype!r}"
    if config is not None:
        merged = {**trainable.model_kwargs, **config}
        d, h = int(merged["d_model"]), int(merged["num_heads"])
        if d % h or d > 256 or d % 4 or (d // h) % 8 or d // h > 64:
            return f"d_model {d} with {h} heads: rows must be <= 256 wide and heads 8..64 wide in steps of 8"
        if int(merged["num_layers"]) > MAX_LAYERS:
            return f"more than {MAX_LAYERS} encoder layers"
    return None


class SeqPackEngine(PackEngine):
    name = "seq"
    move_cost_s = 0.002

    def __init__(self, trainable: PackedTrainable, device: Any = "cuda", capacity: Optional[int] = None, seed: int = 0):
        why = supports(trainable)
        if why:
            raise ValueError(why)
        if not torch.cuda.is_available():
            raise RuntimeError("SeqPackEngine needs a GPU")
        self.lib = L.load()
        _bind(self.lib)
        self.t = trainable
        self.device = dev = torch.device(device)
        self.seed = int(seed or 0)
        kw = trainable.model_kwargs
        self.max_d = int(kw.get("max_d_model", 128))
        self.max_heads = int(kw.get("max_heads", 8))
        self.max_layers = int(kw.get("max_layers", 3))
        self.max_batch = int(kw.get("max_batch", kw.get("batch_size", 32)))
        if self.max_layers > MAX_LAYERS or self.max_d > 256 or self.max_batch > 1024:
            raise ValueError("max_layers <= 8, max_d_model <= 256, max_batch <= 1024")
        self.ksplits = max(1, int(kw.get("ksplits", OPTIONS.seq_ksplits)))
        Xtr, Xva = trainable.train[0], trainable.val[0]
        self.n_train, self.S, self.F = int(Xtr.shape[0]), int(Xtr.shape[1]), int(Xtr.shape[2])
        self.n_val = int(Xva.shape[0])
        self.fpad = ceil4(self.F)
        self.row_len = self.S * self.fpad

        def padded(X: torch.Tensor) -> torch.Tensor:
            out = torch.zeros(X.shape[0], self.S, self.fpad, device=dev, dtype=torch.float32)
            out[:, :, :self.F] = X.to(dev, torch.float32)
            return out.contiguous()

        self.Xtr, self.Xva = padded(Xtr), padded(Xva)
        self.ytr = trainable.train[1].to(dev, torch.float32).contiguous()
        self.yva = trainable.val[1].to(dev, torch.float32).contiguous()
        # Validation runs in chunks of the training batch size, through the training buffers.  Chunks of 128 samples (two
        # passes over a 200-sample validation set instead of seven) were tried: validation got 15 % cheaper, but the
        # forward buffers grew fourfold, spread every slot's working set over four times the address range, and TRAINING
        # slowed by 8 % (64 slots) to 25 % (128 slots).  Compact per-slot regions matter more than fewer launches here.
        self.eval_chunk = int(min(max(1, min(trainable.val_chunk, self.max_batch)), max(self.n_val, 1)))
        self.rows_max = max(self.max_batch, self.eval_chunk)
        self.capacity = int(capacity or trainable.capacity or 64)
        T = self.capacity
        worst = SeqLayout(self.F, self.fpad, self.max_d, self.max_layers)
        self.stride = (worst.total + 63) // 64 * 64
        self.arena_stride = (ArenaLayout(self.rows_max, self.S, self.fpad, self.max_d, self.max_heads,
                                         self.max_layers).total + 63) // 64 * 64
        self.P = torch.zeros(T * self.stride, device=dev, dtype=torch.float32)
        self.G = torch.zeros(self.ksplits * T * self.stride, device=dev, dtype=torch.float32)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.arena = torch.zeros(T * self.arena_stride, device=dev, dtype=torch.float32)
        self.loss_acc = torch.zeros(T, device=dev, dtype=torch.float32)
        self.metrics = torch.zeros(T, 3, device=dev, dtype=torch.float64)
        self.seeds = torch.zeros(T, 2, device=dev, dtype=torch.int32)        # per-slot dropout seed words of the epoch
        self.max_blocks = (self.stride + 2047) // 2048
        self.partials = torch.zeros(T * self.max_blocks, device=dev, dtype=torch.float32)
        self.slots: Dict[int, _SSlot] = {}
        self._stage: Optional[torch.Tensor] = None
        self._tables: Dict[str, Any] = {}
        self._prepared: Dict[int, Tuple[Dict, Any]] = {}          # trial index -> (config, future of its pinned weight image)
        self._init_rng = torch.Generator()
        self._pool = None
        self._pin_pool: List[torch.Tensor] = []          # pinned staging images, recycled once their copy has finished
        self.prepared_total = 0
        self._dirty = True
        self._plans: Dict[str, SeqPlan] = {}
        self._chains: List[SeqPlan] = []
        self._streams: List[torch.cuda.Stream] = []
        self._train_loss: Dict[int, float] = {}
        self._loss_pending = False
        self._round_steps: Dict[int, int] = {}
        self._table = torch.zeros(T + 1, 6, device=dev, dtype=torch.float64)
        self._stamps = torch.zeros(2, device=dev, dtype=torch.int64)

    # ------------------------------------------------------------------ slot management
    def occupied(self) -> List[int]:
        return sorted(self.slots)

    def hbm_bytes(self) -> int:
        ts = [self.P, self.G, self.M, self.V, self.arena, self.Xtr, self.Xva]
        return int(sum(t.numel() * t.element_size() for t in ts))

    def _region(self, slab: torch.Tensor, slot: int) -> torch.Tensor:
        return slab[slot * self.stride:(slot + 1) * self.stride]

    def _spec(self, config: Dict) -> SeqSpec:
        why = supports(self.t, config)
        if why:
            raise ValueError(why)
        spec = seq_spec(self.t, config, self.n_train)
        if spec.d > self.max_d or spec.layers > self.max_layers or spec.heads > self.max_heads:
            raise ValueError(f"trial shape d_model={spec.d} heads={spec.heads} layers={spec.layers} exceeds the pack limits "
                             f"(max_d_model={self.max_d}, max_heads={self.max_heads}, max_layers={self.max_layers})")
        if spec.opt.batch_size > self.max_batch:
            raise ValueError(f"batch_size {spec.opt.batch_size} > max_batch {self.max_batch}")
        return spec

    def _initial_image(self, lay: SeqLayout, trial_index: int, host: torch.Tensor,
                       g: Optional[torch.Generator] = None) -> torch.Tensor:
        """Write the trial's starting weights into ``host`` (a slab-region image): bit for bit those of the module every
        other engine builds for it -- ``SimpleTransformerModel`` under ``manual_seed`` of the trial's stream draws, in
        construction order, a uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) weight and then bias for each dense layer, and sets
        LayerNorm to (1, 0) -- without building the module (a millisecond of host time per trial, which is what bounds a
        search of thousands of one-epoch trials)."""
        host[:lay.total].zero_()
        g = g or self._init_rng         # a private generator: same stream as the global one under the same seed, and
        g.manual_seed((self.seed * 1_000_003 + int(trial_index)) * 16 + 7)        # seeding it costs 1.5 us instead of 160
        bound = 0.0
        for key, off, (rows, cols), ld in lay.entries:       # construction order of the module
            if ".norm" in key:
                if key.endswith(".weight"):
                    host[off:off + cols].fill_(1.0)
                continue
            if key.endswith(".weight"):
                bound = 1.0 / math.sqrt(cols)                # fan_in; the bias that follows shares it
                if ld == cols:
                    host[off:off + rows * cols].view(rows, cols).uniform_(-bound, bound, generator=g)
                else:
                    host[off:off + rows * ld].view(rows, ld)[:, :cols] = torch.empty(rows, cols).uniform_(-bound, bound,
                                                                                                         generator=g)
            else:
                host[off:off + cols].uniform_(-bound, bound, generator=g)
        return host

    def _initial_state(self, spec: SeqSpec, config: Dict, trial_index: int) -> Dict[str, torch.Tensor]:
        """The same weights as a state dict (tests compare it with the module's)."""
        lay = SeqLayout(self.F, self.fpad, spec.d, spec.layers)
        host = torch.zeros(lay.total, dtype=torch.float32)
        self._initial_image(lay, trial_index, host)
        out: Dict[str, torch.Tensor] = {}
        for key, off, (rows, cols), ld in lay.entries:
            t = host[off:off + rows * ld].reshape(rows, ld)[:, :cols].clone()
            out[key] = t.reshape(-1) if (key.endswith(".bias") or ".norm" in key) else t
        return out

    def _pinned(self) -> torch.Tensor:
        return self._pin_pool.pop() if self._pin_pool else torch.zeros(self.stride, dtype=torch.float32).pin_memory()

    def _pack(self, lay: SeqLayout, state: Dict[str, torch.Tensor], host: torch.Tensor) -> None:
        host[:lay.total].zero_()
        for key, off, (rows, cols), ld in lay.entries:
            src = state[key].detach().to("cpu", torch.float32).reshape(rows, cols)
            if ld == cols:
                host[off:off + rows * cols] = src.reshape(-1)
            else:
                host[off:off + rows * ld].view(rows, ld)[:, :cols] = src

    def _write_params(self, slot: int, lay: SeqLayout, state: Dict[str, torch.Tensor], slab: torch.Tensor) -> None:
        if self._stage is None:
            self._stage = torch.zeros(self.stride, dtype=torch.float32).pin_memory()
        self._pack(lay, state, self._stage)
        self._region(slab, slot)[:lay.total].copy_(self._stage[:lay.total])       # blocking: the staging region is free again on return

    def _read_params(self, slot: int, lay: SeqLayout, slab: torch.Tensor) -> Dict[str, torch.Tensor]:
        host = self._region(slab, slot).detach().to("cpu", torch.float32)
        out: Dict[str, torch.Tensor] = {}
        for key, off, (rows, cols), ld in lay.entries:
            t = host[off:off + rows * ld].reshape(rows, ld)[:, :cols].clone()
            out[key] = t.reshape(-1) if (key.endswith(".bias") or ".norm" in key) else t
        return out

    def prepare(self, trial_index, config) -> bool:
        """Build the parameter image of a trial that has no slot yet (host work the executor overlaps with a running
        epoch); ``load`` then only copies it to the device."""
        if int(trial_index) in self._prepared:
            return True
        if len(self._prepared) >= self.capacity:
            return False
        spec = self._spec(config)
        lay = SeqLayout(self.F, self.fpad, spec.d, spec.layers)
        # the CPU generator is the cost (3-6 ns per weight, up to 2 ms per trial): images of the trials to come are drawn
        # by a few worker threads (the tensor library releases the interpreter lock while it fills)
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=max(1, OPTIONS.seq_prep_threads), thread_name_prefix="mitune-seq-init")
        fut = self._pool.submit(self._initial_image, lay, trial_index, self._pinned(), torch.Generator())
        self._prepared[int(trial_index)] = (dict(config), fut)
        self.prepared_total += 1
        return True

    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        spec = self._spec(config)
        s = _SSlot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        s.layout = SeqLayout(self.F, self.fpad, spec.d, spec.layers)
        s.arena = ArenaLayout(self.rows_max, self.S, self.fpad, spec.d, spec.heads, spec.layers)
        s.epoch, s.step = 0, 0
        n = s.layout.total                  # nothing ever touches a region beyond the tenant's own tensors
        for slab in (self.M, self.V):
            self._region(slab, slot)[:n].zero_()
        for sp in range(self.ksplits):      # a previous tenant laid its tensors out differently
            off = (sp * self.capacity + slot) * self.stride
            self.G[off:off + n].zero_()
        ready = self._prepared.pop(int(trial_index), None)
        if state is not None and "model" in state:
            self._write_params(slot, s.layout, state["model"], self.P)
            opt = state.get("optimizer") or {}
            if isinstance(opt, dict) and "exp_avg" in opt and opt.get("type") == spec.opt.optimizer:
                self._write_params(slot, s.layout, opt["exp_avg"], self.M)
                self._write_params(slot, s.layout, opt["exp_avg_sq"], self.V)
            s.epoch, s.step = int(state.get("epoch", 0)), int(state.get("step", 0))
        else:
            if ready is not None and ready[0] == dict(config):
                host = ready[1].result()
            else:
                host = self._pinned()
                self._initial_image(s.layout, trial_index, host)
            self._region(self.P, slot)[:n].copy_(host[:n])        # blocking: the image is free again on return
            self._pin_pool.append(host)
        if ready is not None and (state is not None or ready[0] != dict(config)):
            self._pin_pool.append(ready[1].result())
        self.loss_acc[slot] = 0
        self.slots[slot] = s
        self._dirty = True

    def evict(self, slot):
        if self.slots.pop(slot, None) is not None:
            self._dirty = True

    def export(self, slot):
        s = self.slots[slot]
        return {"model": self._read_params(slot, s.layout, self.P),
                "optimizer": {"type": s.spec.opt.optimizer, "step": s.step,
                              "exp_avg": self._read_params(slot, s.layout, self.M),
                              "exp_avg_sq": self._read_params(slot, s.layout, self.V)},
                "epoch": s.epoch, "step": s.step, "trial_index": s.trial_index}

    def clone(self, src, dst, config):
        a, b = self.slots[src], self.slots[dst]
        spec = self._spec(config)
        if (spec.d, spec.heads, spec.layers) != (a.spec.d, a.spec.heads, a.spec.layers):
            raise ValueError("an exploit cannot change the architecture")
        for slab in (self.P, self.M, self.V):
            self._region(slab, dst).copy_(self._region(slab, src))      # device to device
        b.spec, b.config = spec, dict(config)
        b.layout = SeqLayout(self.F, self.fpad, spec.d, spec.layers)
        b.arena = ArenaLayout(self.rows_max, self.S, self.fpad, spec.d, spec.heads, spec.layers)
        b.step = a.step
        self._dirty = True

    # ------------------------------------------------------------------ moving a trial to another GPU
    @property
    def raw_numel(self) -> int:
        return 3 * self.stride + 4

    def export_raw(self, slot: int) -> torch.Tensor:
        """One slot's parameters and both optimiser moment regions as a single device tensor, epoch / step / trial index
        in the last four words.  A region's layout depends on the trial's shape only, so it is valid in any slot of any
        engine built from the same trainable: another GPU's ``import_raw`` copies it in without the host."""
        s, n = self.slots[slot], self.stride
        out = torch.empty(self.raw_numel, device=self.device, dtype=torch.float32)
        for i, slab in enumerate((self.P, self.M, self.V)):
            out[i * n:(i + 1) * n].copy_(self._region(slab, slot))
        out[3 * n:].view(torch.int32).copy_(torch.tensor([s.epoch, s.step, s.trial_index, 0], dtype=torch.int32))
        return out

    def _take_raw(self, slot: int, raw: torch.Tensor) -> List[int]:
        n = self.stride
        if raw.numel() != self.raw_numel:
            raise ValueError(f"raw slot state of {raw.numel()} words, this engine's slots take {self.raw_numel}")
        raw = raw.to(self.device)
        for i, slab in enumerate((self.P, self.M, self.V)):
            self._region(slab, slot).copy_(raw[i * n:(i + 1) * n])
        for sp in range(self.ksplits):
            off = (sp * self.capacity + slot) * self.stride
            self.G[off:off + self.stride].zero_()
        return raw[3 * n:].view(torch.int32).tolist()

    def import_raw(self, slot: int, trial_index: int, config: Dict, raw: torch.Tensor) -> None:
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        spec = self._spec(config)
        n = self.stride
        owner = int(raw[3 * n:].view(torch.int32)[2])
        if owner != int(trial_index):
            raise ValueError(f"slot state belongs to trial {owner}, not {trial_index}")
        epoch, step, _, _ = self._take_raw(slot, raw)
        s = _SSlot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        s.layout = SeqLayout(self.F, self.fpad, spec.d, spec.layers)
        s.arena = ArenaLayout(self.rows_max, self.S, self.fpad, spec.d, spec.heads, spec.layers)
        s.epoch, s.step = int(epoch), int(step)
        self.loss_acc[slot] = 0
        self.slots[slot] = s
        self._dirty = True

    def adopt_raw(self, slot: int, raw: torch.Tensor, config: Dict) -> None:
        """PBT exploit across GPUs: the occupied ``slot`` takes over a donor's weights, moments and step count (same
        architecture, possibly other hyper-parameters); it keeps its own trial index and epoch count."""
        s = self.slots[slot]
        spec = self._spec(config)
        if (spec.d, spec.heads, spec.layers) != (s.spec.d, s.spec.heads, s.spec.layers):
            raise ValueError("an exploit cannot change the architecture")
        _, step, _, _ = self._take_raw(slot, raw)
        s.spec, s.config, s.step = spec, dict(config), int(step)
        self._dirty = True

    def import_state(self, slot, state, config):
        keep, epoch = self.slots[slot].trial_index, self.slots[slot].epoch
        self.evict(slot)
        self.load(slot, keep, config, state=state)
        self.slots[slot].epoch = epoch

    # ------------------------------------------------------------------ descriptor tables
    def _upload(self, name: str, arr: np.ndarray) -> int:
        host = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        t = self._tables.get(name)
        if isinstance(t, torch.Tensor) and t.numel() == host.numel():
            t.copy_(host)
        else:
            t = host.to(self.device)
            self._tables[name] = t
        return t.data_ptr()

    def _op(self, tag: str, groups: np.ndarray, shapes: List[Tuple[int, int, int, int]], mode2_splits: int = 1) -> SeqOp:
        """Upload one grouped GEMM: ``shapes`` = (group, output rows, output cols, weight) per group.  Always 64 x 64
        tiles: the layers of this family are 32 to 512 wide with contractions of 32 to 512 (the weight gradients: the
        token count), and 128-tiles measured slower at every pack size (16 / 64 / 128 slots: 57 / 38 / 34 us per slot
        step against 68 / 46 / 38)."""
        edge = SMALL_TILE
        tiles = balanced_tiles([(g, (r + edge - 1) // edge, (c + edge - 1) // edge, w) for g, r, c, w in shapes],
                               mode2_splits)
        op = SeqOp()
        op.groups, op.tiles = self._upload(tag + "_g", groups), self._upload(tag + "_t", tiles)
        op.ntiles, op.edge = len(tiles), edge
        return op

    def _build_plan(self, eval_mode: bool, ids: Optional[List[int]] = None, tag: Optional[str] = None) -> SeqPlan:
        T = self.capacity
        tag = tag or ("e" if eval_mode else "t")
        ids = sorted(self.slots) if ids is None else sorted(ids)
        members = [(slot, self.slots[slot]) for slot in ids]
        Lmax = max(s.spec.layers for _, s in members)
        grp = L.gemm_group
        pP, pG, pA = self.P.data_ptr(), self.G.data_ptr(), self.arena.data_ptr()
        seeds = self.seeds.data_ptr()
        S, fp = self.S, self.fpad
        plan = SeqPlan()
        Z = lambda: np.zeros(T, dtype=L.GEMM_GROUP_DTYPE)       # noqa: E731

        info = {}
        for slot, s in members:
            sp = s.spec
            B = self.eval_chunk if eval_mode else min(sp.opt.batch_size, self.n_train)
            steps = 1 if eval_mode else self.n_train // B
            p = 0.0 if eval_mode else sp.dropout
            thresh = min(int(p * 4294967296.0), 0xFFFFFFFF) if p > 0 else 0
            info[slot] = dict(B=B, m=B * S, steps=steps, thresh=thresh, keep=1.0 / (1.0 - p) if p > 0 else 1.0,
                              P=lambda key, s=s, slot=slot: pP + (slot * self.stride + s.layout.off[key]) * 4,
                              G=lambda key, s=s, slot=slot: pG + (slot * self.stride + s.layout.off[key]) * 4,
                              A=lambda key, s=s, slot=slot: pA + (slot * self.arena_stride + s.arena.off[key]) * 4,
                              seed=seeds + slot * 8)

        # ---- input projection
        g_in, sh_in = Z(), []
        for slot, s in members:
            i, d = info[slot], s.spec.d
            a_in = self.Xva.data_ptr() if eval_mode else i["A"]("X0")
            g_in[slot] = grp(a_in, i["P"]("input_projection.weight"), i["A"]("H0"), i["P"]("input_projection.bias"), 0,
                             i["m"], d, fp, fp, fp, d, 0, i["steps"])
            sh_in.append((slot, i["m"], d, 1))
        plan.in_proj = self._op(tag + "in", g_in, sh_in)
        # weight (+ bias) gradients of every dense layer of every slot: ONE launch at the end of the backward pass
        w_groups: List[tuple] = []
        w_shapes: List[Tuple[int, int, int, int]] = []

        def wgrad(rec: tuple, rows: int, cols: int, weight: int) -> None:
            w_shapes.append((len(w_groups), rows, cols, weight))
            w_groups.append(rec)

        if not eval_mode:
            for slot, s in members:
                i, d = info[slot], s.spec.d
                wgrad(grp(i["A"]("dH"), i["A"]("X0"), i["G"]("input_projection.weight"), 0,
                          i["G"]("input_projection.bias"), d, fp, i["m"], d, fp, fp, 0, i["steps"]), d, fp, (i["m"] + 63) // 64)

        # ---- encoder layers
        for l in range(Lmax):
            pre = f"transformer.{l}."
            g_qkv, g_out, g_ff1, g_ff2 = Z(), Z(), Z(), Z()
            sh_qkv, sh_out, sh_ff1, sh_ff2 = [], [], [], []
            dg = {k: Z() for k in ("ff2", "ff1", "out", "qkv")}
            dsh = {k: [] for k in dg}
            attn = np.zeros(T, dtype=ATTN_DTYPE)
            ln1 = np.zeros(T, dtype=LN_DTYPE)
            ln2 = np.zeros(T, dtype=LN_DTYPE)
            for slot, s in members:
                sp = s.spec
                if l >= sp.layers:
                    continue
                i, d, m, st = info[slot], sp.d, info[slot]["m"], info[slot]["steps"]
                A, P, G = i["A"], i["P"], i["G"]
                h_in = A("H0") if l == 0 else A(f"H2_{l - 1}")
                kw_ = (m + 63) // 64
                g_qkv[slot] = grp(h_in, P(pre + "self_attn.qkv_proj.weight"), A(f"QKV{l}"), P(pre + "self_attn.qkv_proj.bias"),
                                  0, m, 3 * d, d, d, d, 3 * d, 0, st)
                sh_qkv.append((slot, m, 3 * d, 1))
                g_out[slot] = grp(A(f"O{l}"), P(pre + "self_attn.out_proj.weight"), A("T"), P(pre + "self_attn.out_proj.bias"),
                                  0, m, d, d, d, d, d, 0, st)
                sh_out.append((slot, m, d, 1))
                g_ff1[slot] = grp(A(f"H1_{l}"), P(pre + "linear1.weight"), A(f"F1_{l}"), P(pre + "linear1.bias"), 0,
                                  m, 4 * d, d, d, d, 4 * d, 0, st, seed=i["seed"], thresh=i["thresh"], scale=i["keep"],
                                  gflags=L.GROUP_ACT, site=4 * l + 2)
                sh_ff1.append((slot, m, 4 * d, 1))
                g_ff2[slot] = grp(A(f"F1_{l}"), P(pre + "linear2.weight"), A("T"), P(pre + "linear2.bias"), 0,
                                  m, d, 4 * d, 4 * d, 4 * d, d, 0, st)
                sh_ff2.append((slot, m, d, 4))
                qkv, dqkv = A(f"QKV{l}"), A(f"dQKV{l}")
                attn[slot] = attn_slot(qkv, qkv + 4 * d, qkv + 8 * d, A(f"O{l}"), A(f"astat{l}"), A("dO"), dqkv, dqkv + 4 * d,
                                       dqkv + 8 * d, i["seed"], i["B"], sp.heads, sp.dk, sp.dk, 3 * d, 3 * d, d, 0, st, 4 * l,
                                       1.0 / math.sqrt(sp.dk), i["keep"], i["thresh"])
                ln1[slot] = (A("T"), h_in, P(pre + "norm1.weight"), P(pre + "norm1.bias"), A(f"H1_{l}"), A(f"Z1_{l}"),
                             A(f"st1_{l}"), A("dH"), A(f"dB1_{l}"), A("dR"), A(f"part1_{l}"), G(pre + "norm1.weight"),
                             G(pre + "norm1.bias"), i["seed"], m, d, st, 4 * l + 1, i["keep"], 1e-5, i["thresh"], 0, 0, 1.0, 0, 0)
                ln2[slot] = (A("T"), A(f"H1_{l}"), P(pre + "norm2.weight"), P(pre + "norm2.bias"), A(f"H2_{l}"),
                             A(f"Z2_{l}"), A(f"st2_{l}"), A("dH"), A(f"dB2_{l}"), A("dR"), A(f"part2_{l}"),
                             G(pre + "norm2.weight"), G(pre + "norm2.bias"), i["seed"], m, d, st, 4 * l + 3, i["keep"],
                             1e-5, i["thresh"], 0, 0, 1.0, 0, 0)
                if eval_mode:
                    continue
                dB2, dF1, dB1, dQKV = A(f"dB2_{l}"), A(f"dF1_{l}"), A(f"dB1_{l}"), A(f"dQKV{l}")
                # feed-forward, second layer: dW2 = dB2^T F1, dF1 = (dB2 W2) * ReLU/dropout mask of F1
                wgrad(grp(dB2, A(f"F1_{l}"), G(pre + "linear2.weight"), 0, G(pre + "linear2.bias"),
                          d, 4 * d, m, d, 4 * d, 4 * d, 0, st), d, 4 * d, kw_)
                dg["ff2"][slot] = grp(dB2, P(pre + "linear2.weight"), dF1, A(f"F1_{l}"), 0,
                                      m, 4 * d, d, d, 4 * d, 4 * d, 4 * d, st, scale=i["keep"], gflags=L.GROUP_ACT)
                dsh["ff2"].append((slot, m, 4 * d, 1))
                # feed-forward, first layer: dW1 = dF1^T H1, dH = dF1 W1 + dR
                wgrad(grp(dF1, A(f"H1_{l}"), G(pre + "linear1.weight"), 0, G(pre + "linear1.bias"),
                          4 * d, d, m, 4 * d, d, d, 0, st), 4 * d, d, kw_)
                dg["ff1"][slot] = grp(dF1, P(pre + "linear1.weight"), A("dH"), A("dR"), 0,
                                      m, d, 4 * d, 4 * d, d, d, d, st, gflags=L.GROUP_ADD)
                dsh["ff1"].append((slot, m, d, 4))
                # attention output projection: dWo = dB1^T O, dO = dB1 Wo
                wgrad(grp(dB1, A(f"O{l}"), G(pre + "self_attn.out_proj.weight"), 0, G(pre + "self_attn.out_proj.bias"),
                          d, d, m, d, d, d, 0, st), d, d, kw_)
                dg["out"][slot] = grp(dB1, P(pre + "self_attn.out_proj.weight"), A("dO"), 0, 0,
                                      m, d, d, d, d, d, 0, st)
                dsh["out"].append((slot, m, d, 1))
                # fused projection: dWqkv = dQKV^T Hin, dH = dQKV Wqkv + dR
                wgrad(grp(dQKV, h_in, G(pre + "self_attn.qkv_proj.weight"), 0, G(pre + "self_attn.qkv_proj.bias"),
                          3 * d, d, m, 3 * d, d, d, 0, st), 3 * d, d, kw_)
                dg["qkv"][slot] = grp(dQKV, P(pre + "self_attn.qkv_proj.weight"), A("dH"), A("dR"), 0,
                                      m, d, 3 * d, 3 * d, d, d, d, st, gflags=L.GROUP_ADD)
                dsh["qkv"].append((slot, m, d, 3))
            lt = f"{tag}{l}"
            plan.qkv[l] = self._op(lt + "qkv", g_qkv, sh_qkv)
            plan.outp[l] = self._op(lt + "out", g_out, sh_out)
            plan.ff1[l] = self._op(lt + "ff1", g_ff1, sh_ff1)
            plan.ff2[l] = self._op(lt + "ff2", g_ff2, sh_ff2)
            plan.attn[l] = self._upload(lt + "attn", attn)
            plan.ln1[l] = self._upload(lt + "ln1", ln1)
            plan.ln2[l] = self._upload(lt + "ln2", ln2)
            if not eval_mode:
                plan.d_ff2[l] = self._op(lt + "dff2", dg["ff2"], dsh["ff2"])
                plan.d_ff1[l] = self._op(lt + "dff1", dg["ff1"], dsh["ff1"])
                plan.d_out[l] = self._op(lt + "dout", dg["out"], dsh["out"])
                plan.d_qkv[l] = self._op(lt + "dqkv", dg["qkv"], dsh["qkv"])
        if not eval_mode:
            wg_all = np.zeros(max(len(w_groups), 1), dtype=L.GEMM_GROUP_DTYPE)
            for k, rec in enumerate(w_groups):
                wg_all[k] = rec
            plan.wgrad = self._op(tag + "wgrad", wg_all, w_shapes, self.ksplits)

        # ---- head
        head = np.zeros(T, dtype=SEQ_HEAD_DTYPE)
        for slot, s in members:
            i = info[slot]
            head[slot] = (i["A"](f"H2_{s.spec.layers - 1}"), i["A"]("dH"), i["P"]("fc.weight"), i["P"]("fc.bias"),
                          i["G"]("fc.weight"), i["G"]("fc.bias"),
                          self.yva.data_ptr() if eval_mode else i["A"]("y"),
                          self.loss_acc.data_ptr() + slot * 4, self.metrics.data_ptr() + slot * 24,
                          i["B"], s.spec.d, LOSS_CODES[s.spec.opt.loss_function], i["steps"])
        plan.head = self._upload(tag + "head", head)
        plan.n_layers, plan.n_active, plan.S = Lmax, len(ids), S
        plan.ids = self._upload(tag + "ids", np.array(ids, dtype=np.int32))
        hmax = max(s.spec.heads for _, s in members)
        if eval_mode:
            plan.max_b, plan.max_bh = self.eval_chunk, self.eval_chunk * hmax
        else:
            plan.max_b = max(info[slot]["B"] for slot in ids)
            plan.max_bh = max(info[slot]["B"] * s.spec.heads for slot, s in members)
        plan.max_m = plan.max_b * S
        plan.max_params = max(s.layout.total for _, s in members)
        plan.wide = max(2, max(int(s.spec.dk) for _, s in members))      # the widest head, in elements
        plan.row_len, plan.hb, plan.n_rows = self.row_len, half_bits(self.n_train), self.n_train
        plan.use_clip = int(any(s.spec.opt.gradient_clipping > 0 for _, s in members))
        plan.ksplits, plan.max_blocks = self.ksplits, self.max_blocks
        plan.X, plan.Y = self.Xtr.data_ptr(), self.ytr.data_ptr()
        plan.P, plan.G, plan.M, plan.V = self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr()
        plan.slot_stride, plan.gsplit_stride = self.stride, self.capacity * self.stride
        plan.partials = self.partials.data_ptr()
        return plan

    def _slot_seed(self, s: _SSlot) -> Tuple[int, int]:
        """Dropout seed words of (experiment seed, trial, epoch): never of the slot or the neighbours."""
        base = np.uint32((self.seed * 1_000_003 + s.trial_index * 4099 + s.epoch * 0x2545F491 + 77) & 0xFFFFFFFF)
        lo = int(mix32(base))
        hi = int(mix32(np.uint32(lo ^ 0x9E3779B9)))
        return lo, hi

    def _epoch_tables(self, plans: List[SeqPlan]) -> int:
        T = self.capacity
        gat = np.zeros(T, dtype=L.GATHER_DTYPE)
        opt = np.zeros(T, dtype=L.OPT_DTYPE)
        seeds = np.zeros((T, 2), dtype=np.uint32)
        max_steps = 0
        pA = self.arena.data_ptr()
        for slot, s in self.slots.items():
            sp = s.spec.opt
            b = min(sp.batch_size, self.n_train)
            steps = self.n_train // b
            max_steps = max(max_steps, steps)
            base = pA + slot * self.arena_stride * 4
            gat[slot] = (base + s.arena.off["X0"] * 4, base + s.arena.off["y"] * 4,
                         round_keys(self.seed, s.trial_index, s.epoch), b, steps)
            opt[slot] = (sp.learning_rate, sp.beta1, sp.beta2, sp.eps, sp.weight_decay, sp.momentum, sp.gradient_clipping,
                         OPT_CODES[sp.optimizer], sp.warmup_steps, sp.total_steps, s.layout.total, steps, s.step, 0, 0, 0)
            seeds[slot] = self._slot_seed(s)
        g_ptr, o_ptr = self._upload("t_gather", gat), self._upload("t_opt", opt)
        for plan in plans:          # the per-epoch tables are indexed by slot: every chain reads the same ones
            plan.gather, plan.opt = g_ptr, o_ptr
        self.seeds.copy_(torch.from_numpy(seeds.view(np.int32)))
        return max_steps

    def _n_chains(self) -> int:
        n = len(self.slots)
        for min_slots, chains in _chain_rule(OPTIONS.seq_chain_rule):
            if n >= min_slots:
                return max(1, min(chains, n))
        return 1

    def _refresh(self) -> None:
        if self._dirty or "e" not in self._plans:
            # independent chains of launches on separate streams: slots dealt round-robin in size order, so every chain
            # gets a similar mix of depths and widths
            n = self._n_chains()
            order = sorted(self.slots, key=lambda k: (-self.slots[k].spec.layers, -self.slots[k].spec.d, k))
            self._chains = [self._build_plan(False, order[c::n], f"t{c}_") for c in range(n)]
            while len(self._streams) < n - 1:
                self._streams.append(torch.cuda.Stream(self.device))
            self._plans["e"] = self._build_plan(True)
            self._dirty = False

    # ------------------------------------------------------------------ training / evaluation
    def train_epoch(self):
        if not self.slots:
            return
        self._refresh()
        steps = self._epoch_tables(self._chains)
        self.loss_acc.zero_()
        main = torch.cuda.current_stream(self.device)
        stream = main.cuda_stream
        if len(self._chains) == 1:
            L.check(self.lib.mt_seq_train_steps(C.byref(self._chains[0]), 0, steps, stream), "mt_seq_train_steps")
        else:
            lanes = [main] + self._streams[:len(self._chains) - 1]
            for st in lanes[1:]:
                st.wait_stream(main)           # tables, seeds and zeroed accumulators were issued on the main stream
            for step in range(steps):          # interleaved step by step: no chain gets a head start
                for plan, st in zip(self._chains, lanes):
                    L.check(self.lib.mt_seq_train_steps(C.byref(plan), step, 1, st.cuda_stream), "mt_seq_train_steps")
            for st in lanes[1:]:
                main.wait_stream(st)
        L.check(self.lib.mt_stamp_launch(self._stamps[1:].data_ptr(), stream), "mt_stamp_launch")
        self._round_steps = {}
        for slot, s in self.slots.items():
            n = self.n_train // min(s.spec.opt.batch_size, self.n_train)
            self._round_steps[slot] = n
            s.step += n
            s.epoch += 1
        self._loss_pending = True

    def begin_round(self) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.mt_stamp_launch(self._stamps.data_ptr(), stream), "mt_stamp_launch")

    def _fetch_losses(self) -> None:
        if not self._loss_pending:
            return
        losses = self.loss_acc.to("cpu")
        for slot, n in self._round_steps.items():
            self._train_loss[slot] = float(losses[slot]) / max(n, 1)
        self._loss_pending = False

    def _enqueue_eval(self) -> None:
        self._refresh()
        self.metrics.zero_()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.mt_seq_eval(C.byref(self._plans["e"]), self.n_val, self.eval_chunk, stream), "mt_seq_eval")

    def device_table(self) -> torch.Tensor:
        """Validate every resident trial and return the round's record table ``[capacity + 1, 6]`` (f64) on the device
        without waiting for it (same layout as ``HipPackEngine.device_table``)."""
        T = self.capacity
        meta = np.zeros(T, dtype=L.TABLE_META_DTYPE)
        for slot, s in self.slots.items():
            meta[slot] = (float(s.trial_index), float(s.epoch), self._round_steps.get(slot, 1), 1)
        if self.slots:
            self._enqueue_eval()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.mt_metric_table_launch(self._upload("table_meta", meta), self.loss_acc.data_ptr(),
                                                self.metrics.data_ptr(), T, max(self.n_val, 1), len(self.slots),
                                                self._stamps.data_ptr(), self._table.data_ptr(), stream),
                "mt_metric_table_launch")
        self._loss_pending = False
        return self._table

    def close(self) -> None:
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def evaluate(self):
        if not self.slots:
            return {}
        self._fetch_losses()
        self._enqueue_eval()
        m = self.metrics.to("cpu")
        out: Dict[int, Dict[str, float]] = {}
        for slot in self.occupied():
            out[slot] = {"train_loss": self._train_loss.get(slot, float("nan")),
                         "validation_loss": float(m[slot, 0]) / self.n_val,
                         "validation_mse": float(m[slot, 1]) / self.n_val,
                         "validation_mape": 100.0 * float(m[slot, 2]) / self.n_val}
        return out
