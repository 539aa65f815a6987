"""HipPackEngine: a pack of MLP trials trained by the hand-written gfx950 kernels.

Memory model (sized for one MI355X, 288 GB of HBM): every slot owns a fixed-stride region of five
flat slabs -- f32 master parameters ``P``, gradients ``G``, optimiser moments ``M``/``V`` and the bf16
shadow ``Pb`` the GEMMs read -- plus per-layer bf16 activation and activation-gradient buffers.
A slot's region is laid out for its *own* architecture (row strides padded to 8 elements so every
operand row starts 16-byte aligned; pad elements are zero and provably stay zero under the update),
so trials of different width and depth share one pack and one launch per layer.

Work submission: once per epoch the engine writes small descriptor tables (GEMM groups + XCD-balanced
tile lists per layer and direction, gather keys, head and optimiser tables) and then makes ONE call
into the C++ driver, which enqueues every step of the epoch back to back.  The host synchronises once
per epoch, when it reads the loss / metric tables.  Compute is bf16 on MFMA with f32 accumulation and
f32 master weights; the optimiser refreshes the bf16 shadow in the same pass.

Behavioural contract shared with ``TorchPackEngine`` (and checked against it in tests): same initial
weights, same keyed-permutation batches, same step count per epoch, same metrics.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import PackEngine
from .permutation import half_bits, round_keys
from . import rowblock_plan
from .trainable import PackedTrainable
from ..models.mlp import MLPTrialSpec, init_mlp_state
from ..ops import lib as L
from ..options import OPTIONS
from ..train.losses import LOSS_CODES
from ..train.optim import OPT_CODES

TILE = 128
N_XCD = 8
MAX_KSPLITS = 4        # weight-gradient slabs kept per slot (split-K over the batch for small packs)
MAX_STREAMS = 4        # concurrent step chains, one HIP stream each
BIG_TILE = 256         # tile edge of the throughput-form GEMM kernels
# 256-tiles (bf16 packs): a launch takes them when it has at least BIG_MIN_TILES of them AND they do not pad much more
# than 128-tiles would: a 256-tile costs BIG_COST 128-tiles (four of them at the eight-wave kernel's better rate), so a
# pack of 512-wide layers qualifies (4 x 3.5 = 14 against 16) and one with many layers of 64-128 columns does not (a
# 256-tile over a 64-column layer is three quarters padding).  Eight-wave kernel on 256 slots x 512^3: forward 110 us
# against 124 on 128-tiles, input gradient 129 / 136, weight gradient 135 / 161 (the earlier four-wave 256-tile form:
# 167 / 309 / 242, which is why this stayed off until the eight-wave form existed).  0 disables.
# (OPTIONS.big_min_tiles, OPTIONS.big_cost; OPTIONS.ksplits_small: weight-gradient splits of packs with <= 128 tiles)
GRAPH_STEPS = 25       # steps per captured hipGraph block (3125 = 125 blocks; shorter epochs run the rest directly)
# Exact-f32 packs: launches with fewer 128-tiles than this use 64x64 tiles instead.  The f32 matrix instruction is 16x
# slower than the bf16 one, so a 64-tile still hides its staging (16 B/clk/CU) and the finer tiles cut both the padding
# of narrow layers and the tail of a launch that does not fill the chip.
# (OPTIONS.f32_small_tiles)
SMALL_TILE = 64
# Launches with fewer 64-tiles than this (a late-round layer of a dozen survivors: about one tile per CU) use 32x32
# tiles: four times the workgroups for the same arithmetic.  Measured on 16 slots x [256 x 128] . [128 x 128]: forward
# 6.5 -> 4.5 us, weight gradient 10.9 -> 6.6 us; at 512 64-tiles the two forms are level.  Headline search by threshold:
# 0 (never) 72.7 k trials/hour, 256 73.9 k, 512 75.5 k, 768 75.1 k, 1024 74.4 k, 2048 74.9 k.
# (OPTIONS.f32_tiny_tiles)
TINY_TILE = 32
# Exact-f32 packs of 32 and more trials run as independent chains of launches on separate HIP streams: every launch of
# the f32 engine ends in a tail of long-running tiles (the f32 matrix instruction is sixteen times slower than the bf16
# one), and a second chain's workgroups fill it.  Packs of a dozen survivors lose from splitting (their launches are
# already too small).  "min_slots:chains,..." -- the first rule whose pack size is reached applies.
def _chain_rule(text: str):
    rule = [tuple(int(v) for v in part.split(":")) for part in text.split(",") if part]
    return sorted(rule, reverse=True)


# (OPTIONS.f32_chain_rule)
# Chains by batch size: a trial's batch size sets how many steps its epoch has (N // batch), so a chain that mixes batch sizes
# runs the step count of its smallest batch with ever fewer live slots in every launch.  With one chain per batch size the
# chains of large batches finish early and the long chain of small batches carries no dead tiles ("0": the mixed deal).
# (OPTIONS.chains_by_batch)
DTYPES = ("bf16", "f32")
# Row-block form of the exact-f32 training step (ops/csrc/rowblock.hip): one launch carries 16 batch rows of a trial
# through every forward layer, the head and every input gradient with the activations staying in LDS, a second one
# computes the weight gradients of all layers, the third is the optimiser -- three launches per step instead of 2 L + 2,
# and no chip-wide dependency between layers.  It is bit-identical to the tile launches, so the engine is free to pick
# per chain: "auto" takes it for chains of at most ROWBLOCK_MAX_BLOCKS 16-row blocks (the late rounds of a search, where a
# step is bound by launch latencies), "1" whenever the shapes allow (every width and the feature count a multiple of 32,
# at most 512), "0" never.
# (OPTIONS.rowblock, OPTIONS.rowblock_max_blocks)
ROWBLOCK_ROWS = 16

def ceil8(x: int) -> int:
    return (x + 7) // 8 * 8


class SlotLayout:
    """Element offsets of one trial's parameters inside its slab region."""

    def __init__(self, spec: MLPTrialSpec, fpad: int):
        self.hidden = list(spec.hidden_sizes)
        self.in_dims = [spec.in_features] + self.hidden[:-1]           # real contraction length per layer
        self.in_ld = [fpad] + [ceil8(h) for h in self.hidden[:-1]]      # row stride of W_l and of its input
        self.out_ld = [ceil8(h) for h in self.hidden]
        off = 0
        self.w_off, self.b_off = [], []
        for h, ld in zip(self.hidden, self.in_ld):
            self.w_off.append(off)
            off += h * ld
            off = ceil8(off)
            self.b_off.append(off)
            off += ceil8(h)
        self.head_w_off = off
        off += self.out_ld[-1]
        self.head_b_off = off
        off += 8
        self.total = off

    def names(self) -> List[Tuple[str, int, Tuple[int, int], int]]:
        """(state_dict key, offset, (rows, cols), row stride) for every parameter tensor."""
        out = []
        for l, h in enumerate(self.hidden):
            out.append((f"layers.{l}.weight", self.w_off[l], (h, self.in_dims[l]), self.in_ld[l]))
            out.append((f"layers.{l}.bias", self.b_off[l], (1, h), h))
        out.append(("head.weight", self.head_w_off, (1, self.hidden[-1]), self.hidden[-1]))
        out.append(("head.bias", self.head_b_off, (1, 1), 1))
        return out


class _HSlot:
    __slots__ = ("trial_index", "spec", "layout", "epoch", "step", "config")


class _Chain:
    """One group of slots trained as an independent chain of launches on its own stream."""
    __slots__ = ("ids", "tag", "plan", "steps", "counter", "graph")


gemm_group = L.gemm_group


def balanced_tiles(groups: List[Tuple[int, int, int, int]], ksplits: int = 1) -> np.ndarray:
    """XCD-aware tile list (``ksplits`` > 1: one entry per contraction split, index in the high byte of ``ti``).

    ``groups`` holds (group index, tiles_i, tiles_j, weight) or (..., weight, splits) for a group with its own number of
    contraction splits.  Workgroup ids are dealt round-robin to
    the 8 XCDs, each with its own L2, so all tiles of one group (which share that group's operands)
    are given ids with equal ``id % 8``: groups go to the currently lightest of 8 buckets, heaviest
    first, and the buckets are interleaved; short buckets are padded with group = -1 entries.
    """
    buckets: List[List[Tuple[int, int, int]]] = [[] for _ in range(N_XCD)]
    load = [0] * N_XCD
    for grp in sorted(groups, key=lambda g: (-g[3] * g[1] * g[2], g[0])):
        gi, ti_n, tj_n, w = grp[:4]
        splits = grp[4] if len(grp) > 4 else ksplits
        b = min(range(N_XCD), key=lambda k: (load[k], k))
        for ti in range(ti_n):
            for tj in range(tj_n):
                for sp in range(splits):
                    buckets[b].append((gi, ti | (sp << 8), tj))
        load[b] += ti_n * tj_n * max(w, 1)
    depth = max((len(b) for b in buckets), default=0)
    out = np.zeros(depth * N_XCD, dtype=L.TILE_DTYPE)
    out["group"] = -1
    for b in range(N_XCD):
        for k, (gi, ti, tj) in enumerate(buckets[b]):
            out[k * N_XCD + b] = (gi, ti, tj)
    return out


class HipPackEngine(PackEngine):
    name = "hip"
    move_cost_s = 0.0005     # three device-to-device region copies and a descriptor refresh (measured 0.12 ms) + the broadcast

    def __init__(self, trainable: PackedTrainable, device: Any = "cuda", capacity: Optional[int] = None,
                 seed: int = 0):
        if trainable.model != "mlp":
            raise ValueError("HipPackEngine trains the 'mlp' family")
        if not torch.cuda.is_available():
            raise RuntimeError("HipPackEngine needs a GPU")
        if trainable.dtype not in DTYPES:
            raise ValueError(f"PackedTrainable.dtype must be one of {DTYPES}, got {trainable.dtype!r}")
        self.lib = L.load()
        self.t = trainable
        self.f32 = trainable.dtype == "f32"
        # element type of activations, activation gradients and of the GEMMs' weight operand: bf16 (MFMA 16x16x32 bf16, f32
        # accumulation, f32 master weights + a bf16 shadow) or f32 end to end (MFMA 16x16x4 f32: the arithmetic of an
        # eager f32 training loop, no shadow)
        self.act_dtype = torch.float32 if self.f32 else torch.bfloat16
        self.eb = 4 if self.f32 else 2
        self.device = torch.device(device)
        self.seed = int(seed or 0)
        kw = trainable.model_kwargs
        self.max_width = int(kw.get("max_width", 512))
        self.max_layers = int(kw.get("max_layers", 3))
        self.max_batch = int(kw.get("max_batch", 1024))
        if self.max_layers > L.MT_MAX_LAYERS or self.max_width > 2048:
            raise ValueError("max_layers <= 8 and max_width <= 2048")
        self.F = trainable.in_features
        self.Fpad = ceil8(self.F)
        dev = self.device
        # ---- datasets: f32 training table (gathered + converted per step), validation table in the compute dtype
        self.Xtr = trainable.train[0].to(dev, torch.float32).contiguous()
        self.ytr = trainable.train[1].to(dev, torch.float32).contiguous()
        xv = torch.zeros(trainable.val[0].shape[0], self.Fpad, device=dev, dtype=self.act_dtype)
        xv[:, :self.F] = trainable.val[0].to(dev, self.act_dtype)
        self.Xva, self.yva = xv, trainable.val[1].to(dev, torch.float32).contiguous()
        self.n_train, self.n_val = int(self.Xtr.shape[0]), int(self.Xva.shape[0])
        self.val_chunk = int(min(trainable.val_chunk, max(self.n_val, 1)))
        self.rows_max = max(self.max_batch, self.val_chunk)
        if capacity or trainable.capacity:
            self.capacity = int(capacity or trainable.capacity)
        else:       # as many slots as the HBM budget allows (288 GB per MI355X): more resident trials amortise the small
            from ..utils.hbm import default_capacity       # late rounds (1024 per GPU measured 1.6x the trials/hour of 256)
            self.capacity = default_capacity(self.F, self.max_width, self.max_layers, self.rows_max, self.n_train,
                                             self.n_val, trainable.hbm_budget_gb, dtype_bytes=self.eb)
        # ---- slabs
        wp = ceil8(self.max_width)
        worst = SlotLayout(MLPTrialSpec(self.F, [self.max_width] * self.max_layers), self.Fpad)
        self.stride = (worst.total + 63) // 64 * 64
        T = self.capacity
        self.P = torch.zeros(T * self.stride, device=dev, dtype=torch.float32)
        self.G = torch.zeros(MAX_KSPLITS * T * self.stride, device=dev, dtype=torch.float32)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.Pb = None if self.f32 else torch.zeros(T * self.stride, device=dev, dtype=torch.bfloat16)
        self.wp = wp
        self.act_stride = self.rows_max * wp
        self.X0 = torch.zeros(T * self.rows_max * self.Fpad, device=dev, dtype=self.act_dtype)
        self.H = [torch.zeros(T * self.act_stride, device=dev, dtype=self.act_dtype) for _ in range(self.max_layers)]
        self.dZ = [torch.zeros(T * self.act_stride, device=dev, dtype=self.act_dtype) for _ in range(self.max_layers)]
        self.ybuf = torch.zeros(T * self.rows_max, device=dev, dtype=torch.float32)
        self.loss_acc = torch.zeros(T, device=dev, dtype=torch.float32)
        self.metrics = torch.zeros(T, 3, device=dev, dtype=torch.float64)
        self.max_blocks = (self.stride + 2047) // 2048
        self.partials = torch.zeros(T * self.max_blocks, device=dev, dtype=torch.float32)
        self.head_scratch = torch.zeros(int(self.lib.mt_head_scratch_floats(T, self.rows_max, self.max_width)),
                                        device=dev, dtype=torch.float32)
        self.slots: Dict[int, _HSlot] = {}
        self._stage: Optional[torch.Tensor] = None
        self._tables: Dict[str, Any] = {}
        self._rb_flags: Dict[str, torch.Tensor] = {}      # chain tag -> exchange flags of its split row blocks
        self._dirty = True
        # Optional concurrent chains: trials are independent, so a pack can be split into up to MAX_STREAMS chains on
        # separate HIP streams.  One host thread cannot feed several chains kernel by kernel (a launch costs ~3.5 us
        # of host time), so each chain's steps are captured in blocks of GRAPH_STEPS as a hipGraph that reads its step
        # number from a device counter and is replayed block after block.  OFF by default ("streams": n or
        # OPTIONS.streams = n enables it): measured wall-clock on the 256-trial search with streams on distinct hardware
        # queues, 7.02 s with one chain, 7.12 s with two or three and 7.23 s with four -- the grouped launches of even a
        # 16-trial pack are wide enough that splitting them only adds launches.  (Kernels of a few workgroups each, as in small Transformer trials, do overlap across
        # streams; packed/engine_graph.py uses that.)
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(MAX_STREAMS)]     # replaced by a calibrated set when chains are on
        self._counters = torch.zeros(MAX_STREAMS, device=dev, dtype=torch.int32)
        self._chain_scratch = [torch.zeros_like(self.head_scratch) for _ in range(MAX_STREAMS - 1)]
        self._chains: List[_Chain] = []
        self.n_streams_override = int(OPTIONS.streams or kw.get("streams", 0) or 0)
        self._train_loss: Dict[int, float] = {}
        self._loss_pending = False          # the epoch's loss accumulators have not been read by the host yet
        self._round_steps: Dict[int, int] = {}
        # the round's record table as the ranks all-gather it, written by one kernel (ops/csrc/runtime.hip)
        self._table = torch.zeros(T + 1, 6, device=dev, dtype=torch.float64)
        self._stamps = torch.zeros(2, device=dev, dtype=torch.int64)      # GPU clock at round start / after training

    # ------------------------------------------------------------------ slot management
    def occupied(self) -> List[int]:
        return sorted(self.slots)

    def hbm_bytes(self) -> int:
        ts = [self.P, self.G, self.M, self.V, self.Pb, self.X0, self.ybuf, self.Xtr, self.Xva] + self.H + self.dZ
        return int(sum(t.numel() * t.element_size() for t in ts if t is not None))

    def _refresh_shadow(self, slot: int) -> None:
        if self.Pb is not None:
            self._region(self.Pb, slot).copy_(self._region(self.P, slot).to(torch.bfloat16))

    def _region(self, slab: torch.Tensor, slot: int) -> torch.Tensor:
        return slab[slot * self.stride:(slot + 1) * self.stride]

    def _write_params(self, slot: int, lay: SlotLayout, state: Dict[str, torch.Tensor], slab: torch.Tensor) -> None:
        host = self._stage                      # one pinned staging region for every upload, not a fresh one per call
        if host is None:
            host = self._stage = torch.zeros(self.stride, dtype=torch.float32).pin_memory()
        else:
            host.zero_()
        for key, off, (rows, cols), ld in lay.names():
            src = state[key].detach().to("cpu", torch.float32).reshape(rows, cols)
            if ld == cols:
                host[off:off + rows * cols] = src.reshape(-1)
            else:       # padded row stride: the pad columns stay zero
                host[off:off + rows * ld].view(rows, ld)[:, :cols] = src
        self._region(slab, slot).copy_(host)      # blocking: the staging region is free again on return

    def _read_params(self, slot: int, lay: SlotLayout, slab: torch.Tensor) -> Dict[str, torch.Tensor]:
        host = self._region(slab, slot).detach().to("cpu", torch.float32)
        out: Dict[str, torch.Tensor] = {}
        for key, off, (rows, cols), ld in lay.names():
            if ld == cols:
                t = host[off:off + rows * cols].reshape(rows, cols).clone()
            else:
                t = host[off:off + rows * ld].reshape(rows, ld)[:, :cols].clone()
            if key.endswith(".bias"):
                t = t.reshape(-1)
            elif key == "head.weight":
                t = t.reshape(1, -1)
            out[key] = t
        return out

    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        spec = MLPTrialSpec.from_config(config, self.F)
        if len(spec.hidden_sizes) > self.max_layers or max(spec.hidden_sizes) > self.max_width:
            raise ValueError(f"trial shape {spec.hidden_sizes} exceeds the pack limits "
                             f"(max_layers={self.max_layers}, max_width={self.max_width})")
        if spec.batch_size > self.max_batch:
            raise ValueError(f"batch_size {spec.batch_size} > max_batch {self.max_batch}")
        s = _HSlot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        s.layout = SlotLayout(spec, self.Fpad)
        s.epoch, s.step = 0, 0
        for slab in (self.M, self.V):
            self._region(slab, slot).zero_()
        for sp in range(MAX_KSPLITS):      # every partial-gradient slab: a previous tenant laid its tensors out differently
            off = (sp * self.capacity + slot) * self.stride
            self.G[off:off + self.stride].zero_()
        if state is not None and "model" in state:
            self._write_params(slot, s.layout, state["model"], self.P)
            opt = state.get("optimizer") or {}
            if isinstance(opt, dict) and "exp_avg" in opt and opt.get("type") == spec.optimizer:
                self._write_params(slot, s.layout, opt["exp_avg"], self.M)
                self._write_params(slot, s.layout, opt["exp_avg_sq"], self.V)
            s.epoch, s.step = int(state.get("epoch", 0)), int(state.get("step", 0))
        else:
            self._write_params(slot, s.layout, init_mlp_state(spec, self.seed, trial_index), self.P)
        self._refresh_shadow(slot)
        self.loss_acc[slot] = 0
        self.slots[slot] = s
        self._dirty = True

    def evict(self, slot):
        if self.slots.pop(slot, None) is not None:
            self._dirty = True

    # measured on one MI355X: the mixed packs of a search sustain about 60 TFLOP/s (f32) / 200 (bf16) of useful arithmetic;
    # on top, a step costs a few microseconds per resident trial (launch boundaries, tails): about 5 us in a pack of 16
    # survivors, much less in a full pack, where the launches are shared by hundreds of trials.  The weights only have
    # to rank trials against each other: arithmetic decides in the early, full rounds, the step count adds to it late.
    COST_FLOPS = {"f32": 6.0e13, "bf16": 2.0e14}
    COST_STEP_S = 2.0e-6

    def trial_cost(self, config) -> float:
        """Seconds one round (a training epoch and a validation pass) of this trial adds to its GPU: arithmetic at the
        rate mixed packs sustain plus a per-step share of the launch chain."""
        spec = MLPTrialSpec.from_config(config, self.F)
        dims = [self.F] + list(spec.hidden_sizes) + [1]
        macs = sum(a * b for a, b in zip(dims[:-1], dims[1:]))
        b = min(spec.batch_size, self.n_train)
        steps = self.n_train // b
        flop = macs * (6.0 * steps * b + 2.0 * self.n_val)
        return flop / self.COST_FLOPS[self.t.dtype] + steps * self.COST_STEP_S

    def export(self, slot):
        s = self.slots[slot]
        return {"model": self._read_params(slot, s.layout, self.P),
                "optimizer": {"type": s.spec.optimizer, "step": s.step,
                              "exp_avg": self._read_params(slot, s.layout, self.M),
                              "exp_avg_sq": self._read_params(slot, s.layout, self.V)},
                "epoch": s.epoch, "step": s.step, "trial_index": s.trial_index}

    def clone(self, src, dst, config):
        a, b = self.slots[src], self.slots[dst]
        spec = MLPTrialSpec.from_config(config, self.F)
        if spec.hidden_sizes != a.spec.hidden_sizes:
            raise ValueError("an exploit cannot change the architecture")
        for slab in (self.P, self.M, self.V, self.Pb):
            if slab is not None:
                self._region(slab, dst).copy_(self._region(slab, src))      # device-to-device, stays in HBM
        b.spec, b.layout, b.config = spec, SlotLayout(spec, self.Fpad), dict(config)
        b.step = a.step
        self._dirty = True

    # ------------------------------------------------------------------ moving a trial to another GPU
    @property
    def raw_numel(self) -> int:
        return 3 * self.stride + 4

    def export_raw(self, slot: int) -> torch.Tensor:
        """One slot's parameters and both optimiser moment regions as a single device tensor, epoch / step counters in
        the last four words.  A slot's layout depends on the trial's shape only, so the regions are valid in any slot
        of any engine built from the same trainable: another GPU's ``import_raw`` copies them in without the host."""
        s, n = self.slots[slot], self.stride
        out = torch.empty(self.raw_numel, device=self.device, dtype=torch.float32)
        for i, slab in enumerate((self.P, self.M, self.V)):
            out[i * n:(i + 1) * n].copy_(self._region(slab, slot))
        meta = torch.tensor([s.epoch, s.step, s.trial_index, 0], dtype=torch.int32)
        out[3 * n:].view(torch.int32).copy_(meta)
        return out

    def import_raw(self, slot: int, trial_index: int, config: Dict, raw: torch.Tensor) -> None:
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        n = self.stride
        if raw.numel() != self.raw_numel:
            raise ValueError(f"raw slot state of {raw.numel()} words, this engine's slots take {self.raw_numel}")
        spec = MLPTrialSpec.from_config(config, self.F)
        epoch, step, owner, _ = raw[3 * n:].view(torch.int32).tolist()
        if owner != int(trial_index):
            raise ValueError(f"slot state belongs to trial {owner}, not {trial_index}")
        s = _HSlot()
        s.trial_index, s.spec, s.config = int(trial_index), spec, dict(config)
        s.layout = SlotLayout(spec, self.Fpad)
        s.epoch, s.step = int(epoch), int(step)
        raw = raw.to(self.device)
        for i, slab in enumerate((self.P, self.M, self.V)):
            self._region(slab, slot).copy_(raw[i * n:(i + 1) * n])
        for sp in range(MAX_KSPLITS):
            off = (sp * self.capacity + slot) * self.stride
            self.G[off:off + self.stride].zero_()
        self._refresh_shadow(slot)
        self.loss_acc[slot] = 0
        self.slots[slot] = s
        self._dirty = True

    def adopt_raw(self, slot: int, raw: torch.Tensor, config: Dict) -> None:
        """PBT exploit across GPUs: the occupied ``slot`` takes over the parameters, both optimiser moments and the step
        count of a donor trial's ``export_raw`` slab (same architecture, possibly other hyper-parameters); it keeps its
        own trial index and epoch count.  Device to device: nothing is unpacked on the host."""
        s, n = self.slots[slot], self.stride
        if raw.numel() != self.raw_numel:
            raise ValueError(f"raw slot state of {raw.numel()} words, this engine's slots take {self.raw_numel}")
        spec = MLPTrialSpec.from_config(config, self.F)
        if spec.hidden_sizes != s.spec.hidden_sizes:
            raise ValueError("an exploit cannot change the architecture")
        raw = raw.to(self.device)
        for i, slab in enumerate((self.P, self.M, self.V)):
            self._region(slab, slot).copy_(raw[i * n:(i + 1) * n])
        self._refresh_shadow(slot)
        _, step, _, _ = raw[3 * n:].view(torch.int32).tolist()
        s.spec, s.layout, s.config, s.step = spec, SlotLayout(spec, self.Fpad), dict(config), int(step)
        self._dirty = True

    def import_state(self, slot, state, config):
        keep = self.slots[slot].trial_index
        epoch = self.slots[slot].epoch
        self.evict(slot)
        self.load(slot, keep, config, state=state)
        self.slots[slot].epoch = epoch

    # ------------------------------------------------------------------ descriptor tables
    def _upload(self, name: str, arr: np.ndarray) -> int:
        """Copy a descriptor table to the device; a table of unchanged size keeps its buffer (and address: captured
        graphs hold these pointers), so refreshing it between epochs is an in-place copy."""
        host = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        t = self._tables.get(name)
        if isinstance(t, torch.Tensor) and t.numel() == host.numel():
            t.copy_(host)
        else:
            t = host.to(self.device)
            self._tables[name] = t
        return t.data_ptr()

    def _act_ptr(self, buf: torch.Tensor, slot: int) -> int:
        return buf.data_ptr() + slot * self.act_stride * self.eb

    def _choose_ksplits(self, ids: Optional[List[int]] = None) -> Tuple[int, int]:
        """(splits for layers >= 1, splits for layer 0) of the wgrad contraction: a launch with too few weight-gradient
        tiles to give every CU one splits the batch dimension instead."""
        per_layer: Dict[int, int] = {}
        edge = SMALL_TILE if self.f32 else TILE      # exact-f32 packs small enough to think of splitting use 64-tiles
        for slot in (ids if ids is not None else sorted(self.slots)):
            lay = self.slots[slot].layout
            for l, h in enumerate(lay.hidden):
                per_layer[l] = per_layer.get(l, 0) + ((h + edge - 1) // edge) * ((lay.in_dims[l] + edge - 1) // edge)

        def rule(tiles: int) -> int:
            # The GEMM alone gains from splitting up to about 128 tiles (512-wide layers: 4 splits 1.4x faster at 128
            # tiles, nothing at 256), but every split is one more f32 slab the weight gradient writes and the optimiser
            # reads, and the optimiser is a quarter of a late-round step: end to end the 256-trial search takes 7.37 s
            # with 4 splits up to 128 tiles, 7.16 s with 2 and 7.15 s with none.  Splits are kept for very small packs.
            if tiles <= 32:
                return min(4, OPTIONS.ksplits_small)
            return min(2, OPTIONS.ksplits_small) if tiles <= 64 else 1

        rest = rule(max([v for l, v in per_layer.items() if l > 0] or [per_layer[0]]))
        return rest, max(rest, rule(per_layer[0]))

    def _build_gemm_tables(self, eval_mode: bool, ids: List[int], tag: str) -> Dict[str, Any]:
        members = [(slot, self.slots[slot]) for slot in ids]
        Lmax = max(len(s.spec.hidden_sizes) for _, s in members)
        ksplits, ksplits0 = (1, 1) if eval_mode else (getattr(self, "_pack_ksplits", None) or self._choose_ksplits(ids))
        self._ksplits, self._ksplits0 = ksplits, ksplits0
        T = self.capacity
        out: Dict[str, Any] = {"n_hidden": Lmax}
        p_base, g_base = self.P.data_ptr(), self.G.data_ptr()
        # what the GEMMs read as the weight operand: the f32 masters themselves, or their bf16 shadow
        w_base, wb = (p_base, 4) if self.f32 else (self.Pb.data_ptr(), 2)
        x0_stride = self.rows_max * self.Fpad * self.eb
        rb_mt = 0 if eval_mode else self._rowblock_mt(ids)
        wg_all = np.zeros(Lmax * T, dtype=L.GEMM_GROUP_DTYPE) if rb_mt else None
        wl_all = []
        for l in range(Lmax):
            fwd = np.zeros(T, dtype=L.GEMM_GROUP_DTYPE)
            dg = np.zeros(T, dtype=L.GEMM_GROUP_DTYPE)
            wg = np.zeros(T, dtype=L.GEMM_GROUP_DTYPE)
            fl, dl, wl = [], [], []            # (slot, rows, cols, weight) per GEMM; tiled below
            for slot, s in members:
                lay = s.layout
                if l >= len(lay.hidden):
                    continue
                bsz = self.val_chunk if eval_mode else s.spec.batch_size
                steps = 1 if eval_mode else self.n_train // min(s.spec.batch_size, self.n_train)
                h, k_real, k_ld, o_ld = lay.hidden[l], lay.in_dims[l], lay.in_ld[l], lay.out_ld[l]
                base = slot * self.stride
                if l == 0:
                    a_in = self.Xva.data_ptr() if eval_mode else self.X0.data_ptr() + slot * x0_stride
                else:
                    a_in = self._act_ptr(self.H[l - 1], slot)
                fwd[slot] = gemm_group(a_in, w_base + (base + lay.w_off[l]) * wb, self._act_ptr(self.H[l], slot),
                                       p_base + (base + lay.b_off[l]) * 4, 0,
                                       bsz, h, k_real, k_ld, k_ld, o_ld, 0, steps)
                fl.append((slot, bsz, h, (k_real + 63) // 64))
                if eval_mode:
                    continue
                if l > 0:
                    dg[slot] = gemm_group(self._act_ptr(self.dZ[l], slot), w_base + (base + lay.w_off[l]) * wb,
                                          self._act_ptr(self.dZ[l - 1], slot), self._act_ptr(self.H[l - 1], slot), 0,
                                          bsz, k_real, h, o_ld, k_ld, k_ld, k_ld, steps)
                    dl.append((slot, bsz, k_real, (h + 63) // 64))
                wg[slot] = gemm_group(self._act_ptr(self.dZ[l], slot), a_in, g_base + (base + lay.w_off[l]) * 4, 0,
                                      g_base + (base + lay.b_off[l]) * 4,
                                      h, k_real, bsz, o_ld, k_ld, k_ld, 0, steps)
                wl.append((slot, h, k_real, (bsz + 63) // 64))
                if rb_mt:      # the same group once more in the table of ALL layers, carrying its layer's split count
                    ks_l = ksplits0 if l == 0 else ksplits
                    wg_all[l * T + slot] = gemm_group(self._act_ptr(self.dZ[l], slot), a_in,
                                                      g_base + (base + lay.w_off[l]) * 4, 0,
                                                      g_base + (base + lay.b_off[l]) * 4,
                                                      h, k_real, bsz, o_ld, k_ld, k_ld, 0, steps, site=ks_l)
                    wl_all.append((l * T + slot, h, k_real, (bsz + 63) // 64, ks_l))
            def tiled(lst, edge):
                return [(g_, (r_ + edge - 1) // edge, (c_ + edge - 1) // edge, w_) for g_, r_, c_, w_ in lst]

            def count(lst, edge):
                return sum(a * b for _, a, b, _ in tiled(lst, edge))

            if self.f32:      # tile-edge codes of the plan: 0 = 128, 2 = 64, 3 = 32
                edge_f = SMALL_TILE if count(fl, TILE) < OPTIONS.f32_small_tiles else TILE
                if count(fl, SMALL_TILE) < OPTIONS.f32_tiny_tiles:
                    edge_f = TINY_TILE
                ks_l = ksplits0 if l == 0 else ksplits
                edge_b = SMALL_TILE if count(dl, TILE) + count(wl, TILE) * ks_l < OPTIONS.f32_small_tiles else TILE
                if count(dl, SMALL_TILE) + count(wl, SMALL_TILE) * ks_l < OPTIONS.f32_tiny_tiles:
                    edge_b = TINY_TILE
                code = {TILE: 0, SMALL_TILE: 2, TINY_TILE: 3}
                out[f"big_fwd{l}"], out[f"big_bwd{l}"] = code[edge_f], code[edge_b]
            else:
                # 256-tiles halve the bytes staged per flop; they pay once a launch still has enough of them to fill the chip
                big_f = (OPTIONS.big_min_tiles > 0 and count(fl, BIG_TILE) >= OPTIONS.big_min_tiles
                         and count(fl, BIG_TILE) * OPTIONS.big_cost <= count(fl, TILE))
                nb_big, nb_128 = count(dl, BIG_TILE) + count(wl, BIG_TILE), count(dl, TILE) + count(wl, TILE)
                big_b = (OPTIONS.big_min_tiles > 0 and (not eval_mode) and ksplits0 == 1 and ksplits == 1
                         and nb_big >= OPTIONS.big_min_tiles and nb_big * OPTIONS.big_cost <= nb_128)
                out[f"big_fwd{l}"], out[f"big_bwd{l}"] = int(big_f), int(big_b)
                edge_f, edge_b = (BIG_TILE if big_f else TILE), (BIG_TILE if big_b else TILE)
            ft = balanced_tiles(tiled(fl, edge_f))
            out[f"fwd{l}"] = (self._upload(f"{tag}fwd_g{l}", fwd), self._upload(f"{tag}fwd_t{l}", ft), len(ft))
            if not eval_mode:
                edge = edge_b
                dt, wt = balanced_tiles(tiled(dl, edge)), balanced_tiles(tiled(wl, edge), ksplits0 if l == 0 else ksplits)
                out[f"dgrad{l}"] = (self._upload(f"{tag}dg_g{l}", dg), self._upload(f"{tag}dg_t{l}", dt), len(dt))
                out[f"wgrad{l}"] = (self._upload(f"{tag}wg_g{l}", wg), self._upload(f"{tag}wg_t{l}", wt), len(wt))
        out["rb_mt"] = rb_mt
        if rb_mt:
            def n_tiles(edge):
                return sum(((r_ + edge - 1) // edge) * ((c_ + edge - 1) // edge) * sp_ for _, r_, c_, _, sp_ in wl_all)
            edge = SMALL_TILE if n_tiles(TILE) < OPTIONS.f32_small_tiles else TILE
            if n_tiles(SMALL_TILE) < OPTIONS.f32_tiny_tiles:
                edge = TINY_TILE
            wt = balanced_tiles([(g_, (r_ + edge - 1) // edge, (c_ + edge - 1) // edge, w_, sp_)
                                 for g_, r_, c_, w_, sp_ in wl_all])
            out["wg_all"] = (self._upload(f"{tag}wg_all_g", wg_all), self._upload(f"{tag}wg_all_t", wt), len(wt),
                             {TILE: 0, SMALL_TILE: 2, TINY_TILE: 3}[edge])
            out["rb_blocks"] = self._rowblock_blocks(ids, rb_mt, tag)
        return out

    def _rowblock_mt(self, ids: List[int]) -> int:
        """Row tiles per workgroup of the row-block step for this chain (1 = 16 rows), or 0 when the chain trains by tile
        launches: a bf16 pack, a shape the kernel does not cover, or (in "auto") too many blocks for the form to pay."""
        if not self.f32 or OPTIONS.rowblock == "0":
            return 0
        wmax = int(self.lib.mt_rowblock_max_width())
        if self.F % 32 or self.F > wmax or self.Fpad != self.F:
            return 0
        blocks = 0
        for slot in ids:
            s = self.slots[slot]
            if any(h % 32 or h > wmax for h in s.layout.hidden) or len(s.layout.hidden) > L.MT_MAX_LAYERS:
                return 0
            blocks += (min(s.spec.batch_size, self.n_train) + ROWBLOCK_ROWS - 1) // ROWBLOCK_ROWS
        if OPTIONS.rowblock != "1" and blocks > OPTIONS.rowblock_max_blocks:
            return 0
        return 1

    def _rb_trials(self, ids: List[int]) -> List["rowblock_plan.RbTrial"]:
        out = []
        for slot in ids:
            s = self.slots[slot]
            nblk = (min(s.spec.batch_size, self.n_train) + ROWBLOCK_ROWS - 1) // ROWBLOCK_ROWS
            out.append(rowblock_plan.RbTrial(slot, nblk, tuple(s.layout.hidden), tuple(s.layout.in_dims)))
        return out

    def _rowblock_split(self, ids: List[int], mt: int) -> Dict[int, int]:
        """Workgroups per 16-row block for every slot of a chain (rowblock_plan.split_parts with this engine's options)."""
        # ranks that share a GPU (more ranks on the node than devices: a rehearsal box) share its CUs too
        n_dev = max(1, torch.cuda.device_count())
        share = max(1, -(-int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1) // n_dev))
        return rowblock_plan.split_parts(self._rb_trials(ids), OPTIONS.rowblock_split if mt == 1 else 1,
                                         OPTIONS.rowblock_split_cost, OPTIONS.rowblock_wg_budget // share)

    def _rowblock_blocks(self, ids: List[int], mt: int, tag: str):
        """The chain's row-block work lists: ((pointer, count) of the list that holds every trial, flags pointer, flag
        count, [(first step, pointer, count) of up to three later lists]).  Trials with larger batches run out of steps
        earlier in the epoch; from each such step on the chain works through a list without them, laid out -- and split
        -- afresh for the trials that are left."""
        def steps_of(slot: int) -> int:
            return self.n_train // min(self.slots[slot].spec.batch_size, self.n_train)

        ends = sorted({steps_of(slot) for slot in ids})[:-1][:3]       # the steps at which somebody drops out
        lists = [self._rowblock_list(ids, mt, f"{tag}rb_blocks")]
        later = []
        for k, first in enumerate(ends):
            left = [slot for slot in ids if steps_of(slot) > first]
            ptr, n, any_split = self._rowblock_list(left, mt, f"{tag}rb_blocks_ph{k}")
            lists.append((ptr, n, any_split))
            later.append((first, ptr, n))
        n_flags = max(n for _, n, _ in lists)
        flags = 0
        self._rb_flags.pop(tag, None)
        if any(sp for _, _, sp in lists):
            t = self._tables.get(f"{tag}rb_flags")
            if not (isinstance(t, torch.Tensor) and t.numel() == n_flags + 1):
                t = torch.zeros(n_flags + 1, dtype=torch.int32, device=self.device)
                self._tables[f"{tag}rb_flags"] = t
            self._rb_flags[tag] = t
            flags = t.data_ptr()
        return lists[0][:2], flags, n_flags, later

    def _rowblock_list(self, ids: List[int], mt: int, name: str) -> Tuple[int, int, bool]:
        """(device pointer, count, any block split) of one work list (rowblock_plan.work_list), uploaded under ``name``."""
        split = self._rowblock_split(ids, mt)
        arr = rowblock_plan.encode(rowblock_plan.work_list(self._rb_trials(ids), split, ROWBLOCK_ROWS), L.RB_BLOCK_DTYPE)
        return self._upload(name, arr), len(arr), any(n > 1 for n in split.values())

    def _build_plan(self, eval_mode: bool, slot_ids: Optional[List[int]] = None, tag: Optional[str] = None,
                    scratch: Optional[torch.Tensor] = None) -> L.MlpPlan:
        T = self.capacity
        slot_ids = sorted(self.slots) if slot_ids is None else sorted(slot_ids)
        members = [(slot, self.slots[slot]) for slot in slot_ids]
        tag = tag or ("e" if eval_mode else "t")
        gt = self._build_gemm_tables(eval_mode, slot_ids, tag)
        plan = L.MlpPlan()
        ids = np.array(slot_ids, dtype=np.int32)
        plan.n_hidden, plan.n_active = gt["n_hidden"], len(ids)
        plan.active_ids = self._upload(tag + "ids", ids)
        plan.head_scratch = (scratch if scratch is not None else self.head_scratch).data_ptr()
        plan.max_blocks = self.max_blocks
        plan.ksplits, plan.gsplit_stride = self._ksplits, self.capacity * self.stride
        plan.ksplits0 = self._ksplits0
        plan.max_batch = max(s.spec.batch_size for _, s in members)
        plan.max_width = max(max(s.spec.hidden_sizes) for _, s in members)
        plan.max_params = max(s.layout.total for _, s in members)
        plan.F, plan.Fpad, plan.hb, plan.n_rows = self.F, self.Fpad, half_bits(self.n_train), self.n_train
        plan.use_clip = int(any(s.spec.gradient_clipping > 0 for _, s in members))
        plan.X, plan.Y = self.Xtr.data_ptr(), self.ytr.data_ptr()
        for l in range(gt["n_hidden"]):
            plan.fwd_groups[l], plan.fwd_tiles[l], plan.fwd_ntiles[l] = gt[f"fwd{l}"]
            plan.fwd_big[l], plan.bwd_big[l] = gt[f"big_fwd{l}"], gt[f"big_bwd{l}"]
            if not eval_mode:
                plan.dgrad_groups[l], plan.dgrad_tiles[l], plan.dgrad_ntiles[l] = gt[f"dgrad{l}"]
                plan.wgrad_groups[l], plan.wgrad_tiles[l], plan.wgrad_ntiles[l] = gt[f"wgrad{l}"]
        if gt.get("rb_mt"):
            plan.rb_mt = gt["rb_mt"]
            (plan.rb_blocks, plan.rb_nblocks), plan.rb_flags, plan.rb_nflags, later = gt["rb_blocks"]
            for k, (first, ptr, n) in enumerate(later):
                plan.rb_ph_from[k], plan.rb_ph_blocks[k], plan.rb_ph_nblocks[k] = first, ptr, n
            plan.wg_all_groups, plan.wg_all_tiles, plan.wg_all_ntiles, plan.wg_all_edge = gt["wg_all"]
        head = np.zeros(T, dtype=L.HEAD_DTYPE)
        for slot, s in members:
            lay, last = s.layout, len(s.layout.hidden) - 1
            base = slot * self.stride
            steps = 1 if eval_mode else self.n_train // min(s.spec.batch_size, self.n_train)
            head[slot] = (self._act_ptr(self.H[last], slot), self._act_ptr(self.dZ[last], slot),
                          self.P.data_ptr() + (base + lay.head_w_off) * 4, self.P.data_ptr() + (base + lay.head_b_off) * 4,
                          self.G.data_ptr() + (base + lay.head_w_off) * 4, self.G.data_ptr() + (base + lay.head_b_off) * 4,
                          self.yva.data_ptr() if eval_mode else self.ybuf.data_ptr() + slot * self.rows_max * 4,
                          self.loss_acc.data_ptr() + slot * 4, self.metrics.data_ptr() + slot * 24,
                          self.val_chunk if eval_mode else s.spec.batch_size, lay.hidden[last], lay.out_ld[last],
                          LOSS_CODES[s.spec.loss_function], steps, 0)
        plan.head = self._upload(tag + "head", head)
        plan.P, plan.G, plan.M, plan.V, plan.Pb = (self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(),
                                                   self.V.data_ptr(), None if self.f32 else self.Pb.data_ptr())
        plan.f32 = int(self.f32)
        plan.slot_stride = self.stride
        plan.partials = self.partials.data_ptr()
        return plan

    def _epoch_tables(self, plan: L.MlpPlan, slot_ids: Optional[List[int]] = None, tag: str = "t") -> int:
        """Gather keys and optimiser step bases change every epoch; returns the chain's step count."""
        T = self.capacity
        gat = np.zeros(T, dtype=L.GATHER_DTYPE)
        opt = np.zeros(T, dtype=L.OPT_DTYPE)
        max_steps = 0
        x0_stride = self.rows_max * self.Fpad * self.eb
        for slot in (sorted(self.slots) if slot_ids is None else slot_ids):
            s = self.slots[slot]
            sp = s.spec
            b = min(sp.batch_size, self.n_train)
            steps = self.n_train // b
            max_steps = max(max_steps, steps)
            gat[slot] = (self.X0.data_ptr() + slot * x0_stride, self.ybuf.data_ptr() + slot * self.rows_max * 4,
                         round_keys(self.seed, s.trial_index, s.epoch), b, steps)
            opt[slot] = (sp.learning_rate, sp.beta1, sp.beta2, sp.eps, sp.weight_decay, sp.momentum,
                         sp.gradient_clipping, OPT_CODES[sp.optimizer], sp.warmup_steps, sp.total_steps,
                         s.layout.total, steps, s.step,
                         s.layout.b_off[0] + ceil8(s.layout.hidden[0]),           # end of layer 0's weights + bias
                         plan.ksplits0 if plan.ksplits0 != plan.ksplits else 0, 0)
        plan.gather = self._upload(tag + "gather", gat)
        plan.opt = self._upload(tag + "opt", opt)
        return max_steps

    # ------------------------------------------------------------------ chains
    def _n_chains(self) -> int:
        if self.n_streams_override:
            return max(1, min(MAX_STREAMS, self.n_streams_override, len(self.slots)))
        if self.f32:
            n = len(self.slots)
            for min_slots, chains in _chain_rule(OPTIONS.f32_chain_rule):       # largest pack size first
                if n >= min_slots:
                    return max(1, min(MAX_STREAMS, chains, n))
        return 1

    def _drop_graphs(self) -> None:
        for ch in self._chains:
            if ch.graph:
                self.lib.mt_mlp_graph_destroy(ch.graph)
                ch.graph = None

    def _partition(self, n: int) -> List[List[int]]:
        """Slot ids per chain.  Mixed deal: round-robin in (batch size, size) order, every chain gets a similar mix and the
        same step count.  By batch size (exact-f32 packs of several batch sizes): one chain per batch size -- merged from
        the small-pack end if there are more batch sizes than streams -- and, while the chain rule asks for more chains,
        the chain with the most steps x slots (the critical path) split in two."""
        order = sorted(self.slots, key=lambda k: (self.slots[k].spec.batch_size, -self.slots[k].layout.total, k))
        by_batch: Dict[int, List[int]] = {}
        for k in order:
            by_batch.setdefault(min(self.slots[k].spec.batch_size, self.n_train), []).append(k)
        # (packs too small for the chain rule stay one chain: 13 + 3 survivors as two chains ran 35 % slower than as one)
        if not (OPTIONS.chains_by_batch and self.f32 and not self.n_streams_override and len(by_batch) > 1 and n > 1):
            return [sorted(order[c::n]) for c in range(n)]
        groups = [by_batch[b] for b in sorted(by_batch)]            # ascending batch size = descending step count
        while len(groups) > MAX_STREAMS:                             # merge the two neighbours with the fewest slots
            i = min(range(len(groups) - 1), key=lambda j: len(groups[j]) + len(groups[j + 1]))
            groups[i:i + 2] = [groups[i] + groups[i + 1]]

        def load(g: List[int]) -> int:
            return len(g) * max(self.n_train // min(self.slots[k].spec.batch_size, self.n_train) for k in g)

        while len(groups) < min(n, MAX_STREAMS):
            i = max(range(len(groups)), key=lambda j: (len(groups[j]) > 1, load(groups[j])))
            if len(groups[i]) < 2:
                break
            groups[i:i + 1] = [groups[i][0::2], groups[i][1::2]]
        return [sorted(g) for g in groups]

    def _check_rb_errors(self) -> None:
        """The word behind a chain's exchange flags is raised by a part of a split row block that gave up waiting for its
        partner; read where the host has synchronised anyway (before the chains are rebuilt, and when the engine closes)."""
        for tag, flags in self._rb_flags.items():
            if int(flags[-1].item()) != 0:
                raise L.KernelLibraryError(f"row-block exchange timed out in chain {tag!r}: a part of a split block never saw "
                                           "its partner's flag, the results of the last epochs are not to be trusted")

    def _build_chains(self) -> None:
        torch.cuda.synchronize(self.device)      # nothing may still be replaying a graph that is about to go
        self._check_rb_errors()
        self._rb_flags = {}
        self._drop_graphs()
        parts = self._partition(self._n_chains())
        n = len(parts)
        self._pack_ksplits = self._choose_ksplits(None)       # from the whole pack: how it is cut into chains changes no arithmetic
        self._chains = []
        for c in range(n):
            ch = _Chain()
            ch.ids = parts[c]
            ch.tag = f"t{c}_"
            ch.plan = self._build_plan(False, ch.ids, ch.tag, None if c == 0 else self._chain_scratch[c - 1])
            ch.steps = 0
            ch.counter = self._counters[c:c + 1]
            ch.graph = None
            self._chains.append(ch)
        self._pack_ksplits = None
        if OPTIONS.chain_debug:
            import sys
            print(f"[mitune] chains: {[(len(ch.ids), sorted({self.slots[k].spec.batch_size for k in ch.ids})) for ch in self._chains]}",
                  file=sys.stderr, flush=True)
        if n > 1:
            from .streams import pick_streams
            self._streams, _ = pick_streams(self.device, MAX_STREAMS)     # streams that land on different hardware queues
            torch.cuda.synchronize(self.device)
            for ch, st in zip(self._chains, self._streams):
                self._epoch_tables(ch.plan, ch.ids, ch.tag)       # gives the per-epoch tables their (stable) addresses
                err = C.c_int(0)
                ch.graph = self.lib.mt_mlp_graph_create(C.byref(ch.plan), GRAPH_STEPS, ch.counter.data_ptr(),
                                                        st.cuda_stream, C.byref(err))
                if not ch.graph:
                    raise L.KernelLibraryError(f"hipGraph capture of a training block failed with code {err.value}")

    def _prepare(self) -> None:
        """(Re)build whatever the current membership of the pack needs: the chains and the validation plan."""
        self._build_chains()
        self._tables["eval_plan"] = self._build_plan(eval_mode=True)
        self._dirty = False

    # ------------------------------------------------------------------ training / evaluation
    def train_epoch(self):
        if not self.slots:
            return
        if self._dirty or not self._chains:
            self._prepare()
        for ch in self._chains:
            ch.steps = self._epoch_tables(ch.plan, ch.ids, ch.tag)
        self.loss_acc.zero_()
        for flags in self._rb_flags.values():
            flags[:-1].zero_()        # split row blocks count their exchanges up from zero through an epoch
        main = torch.cuda.current_stream(self.device)
        if len(self._chains) == 1:
            ch = self._chains[0]
            L.check(self.lib.mt_mlp_train_steps(C.byref(ch.plan), 0, ch.steps, main.cuda_stream), "mt_mlp_train_steps")
        else:
            self._counters.zero_()
            pairs = list(zip(self._chains, self._streams))
            for _, st in pairs:
                st.wait_stream(main)           # tables, counters and zeroed loss accumulators were issued on `main`
            for ch, st in pairs:           # step 0's mini-batch; every later one is gathered by the step before it
                L.check(self.lib.mt_mlp_gather_at(C.byref(ch.plan), ch.counter.data_ptr(), st.cuda_stream),
                        "mt_mlp_gather_at")
            blocks = [ch.steps // GRAPH_STEPS for ch in self._chains]
            # the host stays at most a few blocks ahead of every chain: a host that enqueues graph launches far ahead of
            # several busy streams can wedge the runtime's submission path (packed/engine_graph.py has the measurements)
            from .engine_graph import _FlowControl
            flow = _FlowControl(len(pairs), depth=4)
            for b in range(max(blocks)):       # interleaved so that no chain gets a head start
                for c, ((ch, st), nb) in enumerate(zip(pairs, blocks)):
                    if b < nb:
                        flow.admit(c)
                        L.check(self.lib.mt_mlp_graph_replay(ch.graph, 1, st.cuda_stream), "mt_mlp_graph_replay")
                        flow.sent(c, st)
            for (ch, st), nb in zip(pairs, blocks):
                L.check(self.lib.mt_mlp_train_steps_at(C.byref(ch.plan), ch.counter.data_ptr(),
                                                       ch.steps - nb * GRAPH_STEPS, st.cuda_stream),
                        "mt_mlp_train_steps_at")
                main.wait_stream(st)
        self._finish_epoch(main)

    def _finish_epoch(self, main) -> None:
        L.check(self.lib.mt_stamp_launch(self._stamps[1:].data_ptr(), main.cuda_stream), "mt_stamp_launch")
        self._round_steps = {}
        for slot, s in self.slots.items():
            n = self.n_train // min(s.spec.batch_size, self.n_train)
            self._round_steps[slot] = n
            s.step += n
            s.epoch += 1
        self._loss_pending = True                   # nothing was read back: the host does not wait for the epoch here

    def begin_round(self) -> None:
        """Stamp the GPU clock: the start of a scheduling round (before the epoch is enqueued)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.mt_stamp_launch(self._stamps.data_ptr(), stream), "mt_stamp_launch")

    def _fetch_losses(self) -> None:
        if not self._loss_pending:
            return
        losses = self.loss_acc.to("cpu")            # host synchronisation
        for slot, n in self._round_steps.items():
            self._train_loss[slot] = float(losses[slot]) / max(n, 1)
        self._loss_pending = False

    def _enqueue_eval(self) -> None:
        if self._dirty or "eval_plan" not in self._tables:
            self._prepare()
        plan = self._tables["eval_plan"]
        self.metrics.zero_()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.mt_mlp_eval(C.byref(plan), self.n_val, self.val_chunk, stream), "mt_mlp_eval")

    def device_table(self) -> torch.Tensor:
        """Validate every resident trial and return the round's record table ``[capacity + 1, 6]`` (f64) ON THE DEVICE,
        without waiting for it: rows ``{trial index, epoch, train loss, validation loss, validation MSE, validation
        MAPE}`` per slot (-1 in empty slots), last row ``{-2, seconds since begin_round, live trials, seconds of the
        training part, 0, 0}`` from the GPU clock.  This is the buffer the ranks all-gather."""
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
        self._loss_pending = False        # whoever reads the table has the losses
        return self._table

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

    def close(self) -> None:
        if self._chains:
            torch.cuda.synchronize(self.device)
            self._check_rb_errors()
            self._drop_graphs()
