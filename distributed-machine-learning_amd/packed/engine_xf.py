"""Packs of the configurable sequence-regression Transformer on grouped launches.

The reference's main model family (/root/reference/ray-tune-hpo-regression.py:122-240: input projection, sinusoidal
positions, post-norm encoder layers with softmax or "linear" attention, dense or depthwise-separable feed-forward,
weight sharing, stochastic depth, a five-layer head) over its full search space (:379-400: d_model 64-512, 2-32 heads --
heads 2 to 256 wide, query / key width scaled by 1, 1/2 or 1/4 -- 2-16 layers, dim_feedforward 2-4 x d_model, four losses,
four optimisers, clipping, warm-up / decay) trains here as one PROGRAM per pack: a flat list of grouped launches compiled
from the resident trials' configurations and walked once per optimiser step by ``ops/csrc/xf.hip::mt_xf_run``.  Every
launch covers every resident trial that has the layer in question; what a trial does not have (a convolution, a seventh
layer) is an inactive record in that launch's table.

Arithmetic is f32 throughout.  Dense layers, the head and all weight gradients run on the exact-f32 MFMA tile kernel;
attention with heads of at most 64 elements on the fused exact-f32 kernels; wider heads (or a value width over 64) as
explicit per-head score matrices -- grouped GEMMs for q.k, p.v and their four gradients around a row-normalisation kernel
-- which is exact f32 as well, whatever the width.  Dropout and stochastic depth come from the counter hash of (slot seed
of the epoch, site, step, element), so a trial's numbers depend on neither its slot nor its neighbours.

Weights shared between layers are one parameter block read by every layer's descriptors; each layer writes its gradients
to a block of its own and an ordered sum folds them into the gradient slab before the optimiser.

Contract shared with the other engines (checked against ``TorchPackEngine``): a trial's initial weights are those of
``TransformerModel`` built under the trial's seed, batches come from the keyed permutation of (seed, trial, epoch), an
epoch is ``N // batch`` steps, validation metrics are sample-weighted sums over the whole validation set, and an
exported state loads into the plain module.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import functools
import math
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .engine import PackEngine
from .engine_hip import N_XCD
from .engine_seq import ATTN_DTYPE, LN_DTYPE, attn_slot, ceil4, ceil8
from .permutation import half_bits, mix32, round_keys
from .trainable import PackedTrainable
from ..models.mlp import MLPTrialSpec
from ..ops import lib as L
from ..options import OPTIONS
from ..train.losses import LOSS_CODES
from ..train.optim import OPT_CODES

MAX_LAYERS = 16
HEAD_DIMS = (128, 64, 32, 16)          # fc1 .. fc4 outputs; fc5 maps the last to the prediction
PE_SITE = 4000                         # dropout site of the positional encoding (layer sites are 8 l + k)
FUSED_MAX_HEAD = 64                    # widest head (query / key or value) the fused attention kernels take
GEMM_BF16_PRODUCTS = 0x100      # MT_GEMM_FLAG_BF16_PRODUCTS (csrc/mt_common.h)
MAX_CHAINS = 4                         # concurrent training programs, one HIP stream each
WGRAD_TARGET_TILES = 64                # weight gradients with fewer tiles than this split their contraction up to it

c_void_p, c_int32 = C.c_void_p, C.c_int32


class XfOp(C.Structure):
    """Mirror of ``MtXfOp`` (csrc/xf.hip)."""
    _fields_ = [("kind", c_int32), ("when", c_int32), ("n", c_int32), ("a", c_int32), ("b", c_int32), ("c", c_int32),
                ("d", c_int32), ("flags", c_int32), ("p0", c_void_p), ("p1", c_void_p), ("p2", c_void_p), ("p3", c_void_p)]


class XfOptim(C.Structure):
    _fields_ = [("P", c_void_p), ("G", c_void_p), ("M", c_void_p), ("V", c_void_p), ("partials", c_void_p),
                ("slot_stride", C.c_size_t), ("max_params", c_int32), ("max_blocks", c_int32), ("use_clip", c_int32),
                ("pad", c_int32)]


(K_GATHER, K_GEMM, K_PE_FWD, K_PE_BWD, K_ATTN_FWD, K_ATTN_BWD, K_LN_FWD, K_LN_BWD, K_LN_REDUCE, K_CONV_FWD, K_CONV_BWD,
 K_CONV_WGRAD, K_ROW_FWD, K_ROW_BWD, K_SUM, K_FILL, K_HEAD_LOSS, K_HEAD_EVAL, K_OPTIM) = range(1, 20)

PE_DTYPE = np.dtype([("x", "<u8"), ("pe", "<u8"), ("dx", "<u8"), ("seed", "<u8"), ("M", "<i4"), ("S", "<i4"), ("d", "<i4"),
                     ("active_steps", "<i4"), ("keep_scale", "<f4"), ("thresh", "<u4"), ("site", "<i4"), ("pad", "<i4")],
                    align=True)
CONV_DTYPE = np.dtype([("x", "<u8"), ("w", "<u8"), ("b", "<u8"), ("y", "<u8"), ("dy", "<u8"), ("add", "<u8"), ("dx", "<u8"),
                       ("part", "<u8"), ("gw", "<u8"), ("gb", "<u8"), ("M", "<i4"), ("S", "<i4"), ("d", "<i4"), ("k", "<i4"),
                       ("ldx", "<i4"), ("ldy", "<i4"), ("lddy", "<i4"), ("lddx", "<i4"), ("residual", "<i4"),
                       ("active_steps", "<i4")], align=True)
ROW_DTYPE = np.dtype([("sc", "<u8"), ("pd", "<u8"), ("stat", "<u8"), ("dp", "<u8"), ("seed", "<u8"), ("BH", "<i4"),
                      ("S", "<i4"), ("kind", "<i4"), ("active_steps", "<i4"), ("H", "<i4"), ("site", "<i4"),
                      ("scale", "<f4"), ("keep_scale", "<f4"), ("thresh", "<u4"), ("pad", "<i4")], align=True)
SUM_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("stride", "<i8"), ("n", "<i4"), ("parts", "<i4"),
                      ("active_steps", "<i4"), ("pad", "<i4")], align=True)
FILL_DTYPE = np.dtype([("p", "<u8"), ("n", "<i8"), ("active_steps", "<i4"), ("pad", "<i4")], align=True)

_bound = False


def _bind(lib) -> None:
    global _bound
    if _bound:
        return
    sizes = (C.c_int * 8)()
    lib.mt_xf_sizes(sizes)
    want = [C.sizeof(XfOp), PE_DTYPE.itemsize, CONV_DTYPE.itemsize, ROW_DTYPE.itemsize, SUM_DTYPE.itemsize,
            FILL_DTYPE.itemsize, L.GATHER_DTYPE.itemsize, C.sizeof(XfOptim)]
    if list(sizes) != want:
        raise L.KernelLibraryError(f"xf.hip descriptor sizes {list(sizes)} differ from engine_xf.py {want}")
    if lib.mt_attn_slot_sizeof() != ATTN_DTYPE.itemsize or lib.mt_seq_ln_slot_sizeof() != LN_DTYPE.itemsize:
        raise L.KernelLibraryError("attention / LayerNorm slot layouts differ between the library and engine_seq.py")
    lib.mt_xf_run.argtypes = [c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_void_p]
    lib.mt_xf_run.restype = C.c_int
    _bound = True


# ------------------------------------------------------------------------------------------------------- trial shape
@dataclasses.dataclass
class XfSpec:
    d: int
    heads: int
    layers: int
    dff: int
    dropout: float
    kind: int                # 0 softmax, 1 relu-normalised ("linear") attention
    ksize: int               # local value mixing: kernel size of the depthwise convolution over the tokens, 0 = none
    dk: int
    depthwise: bool
    shared: bool
    sd_rate: float
    opt: MLPTrialSpec

    @property
    def dv(self) -> int:
        return self.d // self.heads

    @property
    def nq(self) -> int:          # columns of the fused projection: all q | all k | all v
        return 2 * self.heads * self.dk + self.d

    @property
    def fused(self) -> bool:
        return self.dk <= FUSED_MAX_HEAD and self.dv <= FUSED_MAX_HEAD

    @property
    def unique_layers(self) -> int:
        return 1 if self.shared else self.layers

    def drop_path(self, layer: int) -> float:
        return self.sd_rate * layer / max(1, self.layers - 1)


def xf_spec(trainable: PackedTrainable, config: Dict, n_train: int) -> XfSpec:
    """The shape ``TransformerModel.from_config`` would build (models/transformer.py) and the trial's optimiser record."""
    merged = {**trainable.model_kwargs, **config}
    opt = MLPTrialSpec.from_config(merged, trainable.in_features)
    if opt.warmup_steps or opt.total_steps:      # same rule as the other engines: the decay never ends before training does
        planned = trainable.max_epochs * max(1, n_train // max(1, min(opt.batch_size, n_train)))
        opt = dataclasses.replace(opt, total_steps=max(opt.total_steps, planned) if opt.total_steps else 0,
                                  warmup_steps=min(opt.warmup_steps, max(1, planned // 4)) if opt.warmup_steps else 0)
    d = int(merged["d_model"])
    heads = int(merged.get("num_heads", 4))
    if d % heads:
        heads = max(h for h in (1, 2, 4, 8, 16, 32) if d % h == 0 and h <= heads)
    dv = d // heads
    kind_name = merged.get("attention_type", "scaled_dot_product")
    linear = kind_name == "linear_attention"
    ks = int(merged.get("attn_kernel_size", 3)) if linear else 0
    return XfSpec(d=d, heads=heads, layers=int(merged.get("num_encoder_layers", merged.get("num_layers", 2))),
                  dff=int(merged.get("dim_feedforward", 4 * d)), dropout=float(merged.get("dropout", 0.1)),
                  kind=1 if linear else 0, ksize=ks if ks > 1 else 0,
                  dk=max(1, int(round(dv * float(merged.get("key_dim_scaling", 1.0))))),
                  depthwise=bool(merged.get("depthwise_separable_conv", False)),
                  shared=bool(merged.get("shared_weights", False)),
                  sd_rate=float(merged.get("stochastic_depth_rate", 0.0)), opt=opt)


def supports(trainable: PackedTrainable, config: Optional[Dict] = None) -> Optional[str]:
    """None if this engine can train the trainable (and, when given, the trial config); else the reason it cannot."""
    if trainable.model != "transformer":
        return "the program engine trains the 'transformer' family"
    X = trainable.train[0]
    if X.dim() != 3:
        return "sequence data [N, S, F] is required"
    if X.shape[1] > 128:
        return "sequences longer than 128 tokens do not fit the attention kernels"
    if trainable.dtype not in ("f32", "bf16"):
        return "the program engine computes in f32, or in f32 with bf16 matrix products ('bf16')"
    if config is not None:
        try:
            sp = xf_spec(trainable, config, int(X.shape[0]))
        except (KeyError, ValueError, TypeError) as e:
            return f"configuration not understood: {e}"
        if sp.d % 4 or sp.d > 512 or sp.dff % 4 or sp.dff > 2048 or sp.layers < 1 or sp.layers > MAX_LAYERS:
            return f"d_model {sp.d} / dim_feedforward {sp.dff} / {sp.layers} layers outside 4..512 / 4..2048 / 1..16"
        if sp.nq % 4:
            return f"{sp.heads} heads of {sp.dk}-wide queries: the fused projection must have a multiple of 4 columns"
        if not sp.fused and (sp.dk % 4 or sp.dv % 4 or X.shape[1] % 4):
            return f"heads {sp.dk} / {sp.dv} wide: heads over 64 elements need widths and a sequence length in multiples of 4"
        if sp.ksize not in (0, 3, 5, 7):
            return f"attn_kernel_size {sp.ksize}: local mixing takes odd kernels up to 7"
    return None


class XfLayout:
    """Element offsets of one trial's parameter tensors inside its slab region, keyed like ``TransformerModel``'s
    ``state_dict``.  Every tensor starts on a multiple of 8 floats; the input projection's rows are padded to ``fpad``
    columns.  ``layer_lo[l] .. layer_hi[l]`` is the block of unique layer ``l`` (one block when weights are shared)."""

    def __init__(self, F: int, fpad: int, sp: XfSpec):
        self.entries: List[Tuple[str, int, Tuple[int, ...], int, int, int]] = []   # key, offset, shape, rows, cols, row stride
        self.off: Dict[str, int] = {}
        at = 0

        def add(key: str, shape: Tuple[int, ...], ld: Optional[int] = None):
            nonlocal at
            rows = shape[0] if len(shape) > 1 else 1
            cols = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            ld = ld or cols
            self.entries.append((key, at, shape, rows, cols, ld))
            self.off[key] = at
            at = ceil8(at + rows * ld)

        d, h, dff = sp.d, sp.heads, sp.dff
        add("input_projection.weight", (d, F), fpad)
        add("input_projection.bias", (d,))
        self.layer_lo, self.layer_hi = [], []
        for l in range(sp.unique_layers):
            p = f"layers.{l}."
            self.layer_lo.append(at)
            add(p + "attention.qkv_proj.weight", (sp.nq, d))
            add(p + "attention.qkv_proj.bias", (sp.nq,))
            add(p + "attention.out_proj.weight", (d, d))
            add(p + "attention.out_proj.bias", (d,))
            if sp.ksize:
                add(p + "attention.local.weight", (d, 1, sp.ksize))
                add(p + "attention.local.bias", (d,))
            if sp.depthwise:
                add(p + "feedforward.depthwise.weight", (d, 1, 3))
                add(p + "feedforward.depthwise.bias", (d,))
                add(p + "feedforward.pointwise.weight", (dff, d, 1))
                add(p + "feedforward.pointwise.bias", (dff,))
                add(p + "feedforward.project.weight", (d, dff, 1))
                add(p + "feedforward.project.bias", (d,))
            else:
                add(p + "feedforward.0.weight", (dff, d))
                add(p + "feedforward.0.bias", (dff,))
                add(p + "feedforward.2.weight", (d, dff))
                add(p + "feedforward.2.bias", (d,))
            for n in ("norm1", "norm2"):
                add(p + n + ".weight", (d,))
                add(p + n + ".bias", (d,))
            self.layer_hi.append(at)
        k = d
        for i, o in enumerate(HEAD_DIMS + (1,)):
            add(f"fc{i + 1}.weight", (o, k))
            add(f"fc{i + 1}.bias", (o,))
            k = o
        self.total = at
        self.layer_span = (self.layer_hi[0] - self.layer_lo[0]) if self.layer_lo else 0


class XfArena:
    """Float offsets of one slot's activation and gradient buffers for ``rows`` samples of ``S`` tokens."""

    def __init__(self, rows: int, S: int, fpad: int, sp: XfSpec, layout: XfLayout):
        m, d, h = rows * S, sp.d, sp.heads
        self.off: Dict[str, int] = {}
        at = 0

        def add(key: str, n: int):
            nonlocal at
            self.off[key] = at
            at = ceil4(at + n)

        add("X0", m * fpad)
        add("y", rows)
        add("pe", S * d)
        add("H0", m * d)
        add("T", m * d)
        scores = rows * h * S * S
        for l in range(sp.layers):
            add(f"QKV{l}", m * sp.nq)
            if sp.ksize:
                add(f"VM{l}", m * d)
            add(f"O{l}", m * d)
            add(f"Z1_{l}", m * d)
            add(f"H1_{l}", m * d)
            if sp.depthwise:
                add(f"C1_{l}", m * d)
            add(f"F1_{l}", m * sp.dff)
            add(f"Z2_{l}", m * d)
            add(f"H2_{l}", m * d)
            add(f"st1_{l}", 2 * m)
            add(f"st2_{l}", 2 * m)
            add(f"astat{l}", rows * h * S)
            if not sp.fused:
                add(f"Pn{l}", scores)
                if sp.dropout > 0:
                    add(f"Pd{l}", scores)
            add(f"part1_{l}", ((m + 31) // 32) * 2 * d)
            add(f"part2_{l}", ((m + 31) // 32) * 2 * d)
        for key, width in (("dH", d), ("dR", d), ("dO", d), ("dB1", d), ("dB2", d), ("dF1", sp.dff), ("dQKV", sp.nq)):
            add(key, m * width)
        if sp.ksize:
            add("dVM", m * d)
            add("cpartA", ((m + 63) // 64) * (sp.ksize + 1) * d)
        if sp.depthwise:
            add("dC1", m * d)
            add("cpartF", ((m + 63) // 64) * 4 * d)
        if not sp.fused:
            add("dP", scores)
        for i, o in enumerate(HEAD_DIMS):
            add(f"A{i + 1}", rows * o)
            add(f"dA{i + 1}", rows * o)
        if sp.shared:
            add("GL", sp.layers * layout.layer_span)
        self.total = at


class _XSlot:
    __slots__ = ("trial_index", "spec", "layout", "arena", "buf", "epoch", "step", "config")


def _uniform_tiles(n_groups: int, ti: int, tj: int) -> np.ndarray:
    """Tile list of ``n_groups`` equal GEMMs of ``ti`` x ``tj`` tiles, group g on XCD g % 8 (vectorised: the per-head
    score products of one wide-head trial are thousands of groups)."""
    per = ti * tj
    depth_g = (n_groups + N_XCD - 1) // N_XCD
    out = np.zeros((depth_g * per, N_XCD), dtype=L.TILE_DTYPE)
    out["group"] = -1
    g = np.arange(depth_g * N_XCD).reshape(depth_g, N_XCD)                       # group in (row block, bucket)
    g = np.repeat(g, per, axis=0)
    t = np.tile(np.arange(per), depth_g)[:, None]
    ok = g < n_groups
    out["group"][ok] = g[ok]
    out["ti"] = (t // tj).astype(np.int16)
    out["tj"] = (t % tj).astype(np.int16)
    out["group"][~ok] = -1
    return out.reshape(-1)


def spread_tiles(shapes: List[Tuple[int, ...]]) -> np.ndarray:
    """``_spread_tiles`` through a cache: the launches of a program repeat a few shape lists layer after layer, and most of
    them survive a change of the pack (building the lists was 1.1 s of the 8.7 s full-workload search).  The result is
    shared: read only."""
    return _spread_tiles_cached(tuple(tuple(int(v) for v in x) for x in shapes))


@functools.lru_cache(maxsize=8192)
def _spread_tiles_cached(shapes: Tuple[Tuple[int, ...], ...]) -> np.ndarray:
    out = _spread_tiles(list(shapes))
    out.setflags(write=False)
    return out


def _spread_tiles(shapes: List[Tuple[int, ...]]) -> np.ndarray:
    """Tile list of a launch with few, very unequal groups (a pack of a dozen Transformer trials whose token counts and
    widths differ by orders of magnitude): ``shapes`` = (group, row tiles, column tiles[, contraction splits: one entry
    per split, index in the high byte of ``ti``]).  Workgroup ids go round-robin
    to the 8 XCDs, each with its own L2; here every group is cut into bands of row tiles and band r goes to XCD
    (start + r) % 8, ``start`` being the lightest XCD when the group is placed (largest groups first).  All XCDs work on a
    large group at once -- its activation rows are private to a band, its weights (the operand every band shares) are
    small enough to sit in each L2 -- where one-group-per-XCD placement would leave seven eighths of the chip idle
    behind the largest trial."""
    load = np.zeros(N_XCD, dtype=np.int64)
    per_bucket: List[List[np.ndarray]] = [[] for _ in range(N_XCD)]
    for shape in sorted(shapes, key=lambda x: (-x[1] * x[2], x[0])):
        g, ti_n, tj_n = shape[:3]
        splits = shape[3] if len(shape) > 3 else 1
        if ti_n <= 0 or tj_n <= 0:
            continue
        start = int(np.argmin(load))
        ti = np.repeat(np.arange(ti_n, dtype=np.int64), tj_n * splits)
        tj = np.tile(np.repeat(np.arange(tj_n, dtype=np.int64), splits), ti_n)
        sp = np.tile(np.arange(splits, dtype=np.int64), ti_n * tj_n)
        bucket = (start + ti + sp) % N_XCD if splits > 1 else (start + ti) % N_XCD
        ti = ti | (sp << 8)
        for b in range(N_XCD):
            sel = bucket == b
            k = int(sel.sum())
            if k:
                rec = np.zeros(k, dtype=L.TILE_DTYPE)
                rec["group"], rec["ti"], rec["tj"] = g, ti[sel], tj[sel]
                per_bucket[b].append(rec)
                load[b] += k
    depth = int(load.max()) if len(shapes) else 0
    out = np.zeros((depth, N_XCD), dtype=L.TILE_DTYPE)
    out["group"] = -1
    for b in range(N_XCD):
        if per_bucket[b]:
            col = np.concatenate(per_bucket[b])
            out[:len(col), b] = col
    return out.reshape(-1)


class _ArenaFull(Exception):
    pass


class _TableArena:
    """The descriptor tables of one set of programs as ONE device buffer filled by ONE copy from a pinned host image.  A
    16-layer pack has some 2 500 tables (group records, tile lists, per-slot records); uploaded one by one they were 0.2 s
    of host time per change of the pack, with the GPU idle."""

    def __init__(self, device: torch.device, capacity: int):
        self.capacity = int(capacity)
        self.host = torch.empty(self.capacity, dtype=torch.uint8).pin_memory()
        self.image = self.host.numpy()
        self.dev = torch.empty(self.capacity, dtype=torch.uint8, device=device)
        self.used = 0
        self.index: Dict[str, Tuple[int, int]] = {}
        self.copied: Optional[torch.cuda.Event] = None

    def reset(self) -> None:
        if self.copied is not None:
            self.copied.synchronize()        # the last copy out of the host image has been made
        self.used, self.index = 0, {}

    def put(self, name: str, arr: np.ndarray) -> int:
        raw = arr.view(np.uint8).reshape(-1)
        n = int(raw.size)
        off = (self.used + 63) // 64 * 64
        if off + n > self.capacity:
            raise _ArenaFull()
        self.image[off:off + n] = raw
        self.index[name] = (off, n)
        self.used = off + n
        return self.dev.data_ptr() + off

    def flush(self) -> None:
        if self.used:
            self.dev[:self.used].copy_(self.host[:self.used], non_blocking=True)
            self.copied = torch.cuda.Event()
            self.copied.record()

    def update(self, name: str, arr: np.ndarray) -> int:
        """Replace one table after the flush (the per-epoch tables): a small copy of its own, from a fresh host tensor."""
        off, n = self.index[name]
        raw = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        if raw.numel() != n:
            raise ValueError(f"table {name!r} changed size")
        self.dev[off:off + n].copy_(raw)
        return self.dev.data_ptr() + off


class XfPackEngine(PackEngine):
    name = "xf"
    move_cost_s = 0.01
    prepare_is_cheap = True      # prepare() draws weights on the host: the executor may prepare for every slot that may free up

    def __init__(self, trainable: PackedTrainable, device: Any = "cuda", capacity: Optional[int] = None, seed: int = 0):
        why = supports(trainable)
        if why:
            raise ValueError(why)
        if not torch.cuda.is_available():
            raise RuntimeError("XfPackEngine needs a GPU")
        self.lib = L.load()
        _bind(self.lib)
        self.t = trainable
        self.device = dev = torch.device(device)
        self.seed = int(seed or 0)
        # dtype "bf16": parameters, activations, accumulation and every non-matrix kernel stay f32; the grouped GEMMs round
        # their operands to bf16 for the matrix instruction (the arithmetic of bf16 autocast, with f32 outputs)
        self.product_flag = GEMM_BF16_PRODUCTS if trainable.dtype == "bf16" else 0
        kw = trainable.model_kwargs
        self.max_d = int(kw.get("max_d_model", 512))
        self.max_layers = int(kw.get("max_layers", MAX_LAYERS))
        self.max_dff = int(kw.get("max_dim_feedforward", 4 * self.max_d))
        self.max_batch = int(kw.get("max_batch", kw.get("batch_size", 256)))
        if self.max_layers > MAX_LAYERS or self.max_d > 512 or self.max_batch > 1024 or self.max_dff > 2048:
            raise ValueError("max_layers <= 16, max_d_model <= 512, max_dim_feedforward <= 2048, max_batch <= 1024")
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
        self.eval_chunk = int(min(max(1, min(trainable.val_chunk, self.max_batch)), max(self.n_val, 1)))
        self.capacity = int(capacity or trainable.capacity or 16)
        T = self.capacity
        worst = XfSpec(d=self.max_d, heads=2, layers=self.max_layers, dff=self.max_dff, dropout=0.1, kind=1, ksize=7,
                       dk=self.max_d // 2, depthwise=False, shared=False, sd_rate=0.0,
                       opt=MLPTrialSpec.from_config({"batch_size": self.max_batch}, self.F))
        dense = XfLayout(self.F, self.fpad, worst).total
        worst.depthwise = True
        self.stride = (max(dense, XfLayout(self.F, self.fpad, worst).total) + 63) // 64 * 64
        self.P = torch.zeros(T * self.stride, device=dev, dtype=torch.float32)
        self.G = torch.zeros(T * self.stride, device=dev, dtype=torch.float32)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.loss_acc = torch.zeros(T, device=dev, dtype=torch.float32)
        self.metrics = torch.zeros(T, 3, device=dev, dtype=torch.float64)
        self.seeds = torch.zeros(T, 2, device=dev, dtype=torch.int32)
        self.max_blocks = (self.stride + 2047) // 2048
        self.partials = torch.zeros(T * self.max_blocks, device=dev, dtype=torch.float32)
        rows_max = max(self.max_batch, self.eval_chunk)
        self.head_scratch = torch.zeros(int(self.lib.mt_head_scratch_floats(T, rows_max, HEAD_DIMS[-1])), device=dev)
        self.slots: Dict[int, _XSlot] = {}
        self._tables: Dict[str, Any] = {}
        self._arenas: List[Optional[_TableArena]] = [None, None]      # the programs in force and the ones before them
        self._arena: Optional[_TableArena] = None
        self._arena_open = False
        self._arena_turn = 0
        self._stage: Optional[torch.Tensor] = None
        self._dirty = True
        self._chains: List[Dict[str, Any]] = []
        self._eval: Dict[str, Any] = {}
        self._streams: List[torch.cuda.Stream] = []
        self._train_loss: Dict[int, float] = {}
        self._loss_pending = False
        self._round_steps: Dict[int, int] = {}
        self._table = torch.zeros(T + 1, 6, device=dev, dtype=torch.float64)
        self._stamps = torch.zeros(2, device=dev, dtype=torch.int64)
        self.prepared_total = 0
        self._prepared: Dict[int, Tuple[Dict, torch.Tensor]] = {}      # trial index -> (config, parameter image on the host)
        self._meta_host: Optional[torch.Tensor] = None
        self._meta_copied: List[Optional[torch.cuda.Event]] = [None, None]
        self._meta_turn = 0

    # ------------------------------------------------------------------ slot management
    def occupied(self) -> List[int]:
        return sorted(self.slots)

    def hbm_bytes(self) -> int:
        ts = [self.P, self.G, self.M, self.V, self.Xtr, self.Xva] + [s.buf for s in self.slots.values()]
        return int(sum(t.numel() * t.element_size() for t in ts))

    def trial_cost(self, config: Dict) -> float:
        sp = xf_spec(self.t, config, self.n_train)
        per_token = sp.layers * (sp.d * sp.nq + sp.d * sp.d + 2 * sp.d * sp.dff + 2 * self.S * (sp.heads * sp.dk + sp.d))
        return float(per_token) * self.S * self.n_train / 1e9

    def _region(self, slab: torch.Tensor, slot: int) -> torch.Tensor:
        return slab[slot * self.stride:(slot + 1) * self.stride]

    def _spec(self, config: Dict) -> XfSpec:
        why = supports(self.t, config)
        if why:
            raise ValueError(why)
        sp = xf_spec(self.t, config, self.n_train)
        if sp.d > self.max_d or sp.layers > self.max_layers or sp.dff > self.max_dff or sp.opt.batch_size > self.max_batch:
            raise ValueError(f"trial shape d_model={sp.d} layers={sp.layers} dim_feedforward={sp.dff} batch={sp.opt.batch_size} "
                             f"exceeds the pack limits (max_d_model={self.max_d}, max_layers={self.max_layers}, "
                             f"max_dim_feedforward={self.max_dff}, max_batch={self.max_batch})")
        return sp

    def _initial_state(self, config: Dict, trial_index: int) -> Dict[str, torch.Tensor]:
        """The weights every other engine starts this trial from: the module built under the trial's seed."""
        from ..models.transformer import TransformerModel
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed((self.seed * 1_000_003 + int(trial_index)) * 16 + 7)
            model = TransformerModel.from_config({**self.t.model_kwargs, **config}, self.F)
        return {k: v.detach() for k, v in model.state_dict().items()}

    def _pack(self, lay: XfLayout, state: Dict[str, torch.Tensor], host: torch.Tensor) -> None:
        host[:lay.total].zero_()
        for key, off, shape, rows, cols, ld in lay.entries:
            src = state[key].detach().to("cpu", torch.float32).reshape(rows, cols)
            if ld == cols:
                host[off:off + rows * cols] = src.reshape(-1)
            else:
                host[off:off + rows * ld].view(rows, ld)[:, :cols] = src

    def _write_params(self, slot: int, lay: XfLayout, state: Dict[str, torch.Tensor], slab: torch.Tensor) -> None:
        if self._stage is None:
            self._stage = torch.zeros(self.stride, dtype=torch.float32).pin_memory()
        self._pack(lay, state, self._stage)
        self._region(slab, slot)[:lay.total].copy_(self._stage[:lay.total])

    def _read_params(self, slot: int, lay: XfLayout, slab: torch.Tensor) -> Dict[str, torch.Tensor]:
        host = self._region(slab, slot)[:lay.total].detach().to("cpu", torch.float32)
        out: Dict[str, torch.Tensor] = {}
        for key, off, shape, rows, cols, ld in lay.entries:
            out[key] = host[off:off + rows * ld].reshape(rows, ld)[:, :cols].clone().reshape(shape)
        return out

    def _make_slot(self, trial_index: int, config: Dict) -> _XSlot:
        s = _XSlot()
        s.trial_index, s.spec, s.config = int(trial_index), self._spec(config), dict(config)
        s.layout = XfLayout(self.F, self.fpad, s.spec)
        rows = max(min(s.spec.opt.batch_size, self.n_train), self.eval_chunk)
        s.arena = XfArena(rows, self.S, self.fpad, s.spec, s.layout)
        s.buf = torch.zeros(s.arena.total, device=self.device, dtype=torch.float32)
        from ..models.transformer import sinusoid_table
        o = s.arena.off["pe"]
        s.buf[o:o + self.S * s.spec.d].copy_(sinusoid_table(self.S, s.spec.d).reshape(-1))
        s.epoch, s.step = 0, 0
        return s

    def prepare(self, trial_index, config) -> bool:
        """Draw a coming trial's initial weights and lay them out as its parameter image on the host (the executor calls
        this once an epoch and its validation are enqueued: the GPU is still busy); ``load`` then only copies the image.
        Returns False when the cache is full."""
        key = int(trial_index)
        if key in self._prepared:
            return True
        if len(self._prepared) >= self.capacity:
            return False
        spec = self._spec(config)
        lay = XfLayout(self.F, self.fpad, spec)
        image = torch.zeros(lay.total, dtype=torch.float32)
        self._pack(lay, self._initial_state(config, trial_index), image)
        self._prepared[key] = (dict(config), image)
        self.prepared_total += 1
        return True

    def load(self, slot, trial_index, config, state=None):
        if slot in self.slots:
            raise RuntimeError(f"slot {slot} is occupied")
        s = self._make_slot(trial_index, config)
        n = s.layout.total
        for slab in (self.M, self.V, self.G):
            self._region(slab, slot)[:n].zero_()
        ready = self._prepared.pop(int(trial_index), None)
        if ready is not None and state is None and ready[0] == dict(config):
            self._region(self.P, slot)[:n].copy_(ready[1][:n])
            self.loss_acc[slot] = 0
            self.slots[slot] = s
            self._dirty = True
            return
        if state is not None and "model" in state:
            self._write_params(slot, s.layout, state["model"], self.P)
            opt = state.get("optimizer") or {}
            if isinstance(opt, dict) and "exp_avg" in opt and opt.get("type") == s.spec.opt.optimizer:
                self._write_params(slot, s.layout, opt["exp_avg"], self.M)
                self._write_params(slot, s.layout, opt["exp_avg_sq"], self.V)
            s.epoch, s.step = int(state.get("epoch", 0)), int(state.get("step", 0))
        else:
            self._write_params(slot, s.layout, self._initial_state(config, trial_index), self.P)
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

    @staticmethod
    def _same_shape(a: XfSpec, b: XfSpec) -> bool:
        keys = ("d", "heads", "layers", "dff", "kind", "ksize", "dk", "depthwise", "shared")
        return all(getattr(a, k) == getattr(b, k) for k in keys)

    def clone(self, src, dst, config):
        a, b = self.slots[src], self.slots[dst]
        spec = self._spec(config)
        if not self._same_shape(spec, a.spec):
            raise ValueError("an exploit cannot change the architecture")
        for slab in (self.P, self.M, self.V):
            self._region(slab, dst).copy_(self._region(slab, src))      # device to device
        keep_index, keep_epoch = b.trial_index, b.epoch
        nb = self._make_slot(keep_index, config)
        nb.epoch, nb.step = keep_epoch, a.step
        self.slots[dst] = nb
        self._dirty = True

    def import_state(self, slot, state, config):
        keep, epoch = self.slots[slot].trial_index, self.slots[slot].epoch
        self.evict(slot)
        self.load(slot, keep, config, state=state)
        self.slots[slot].epoch = epoch

    # ------------------------------------------------------------------ descriptor tables
    def _upload(self, name: str, arr: np.ndarray) -> int:
        if self._arena is not None:
            if self._arena_open:
                return self._arena.put(name, arr)
            if name in self._arena.index:
                return self._arena.update(name, arr)
        host = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        t = self._tables.get(name)
        if isinstance(t, torch.Tensor) and t.numel() == host.numel():
            t.copy_(host)
        else:
            t = host.to(self.device)
            self._tables[name] = t
        return t.data_ptr()

    def _slot_seed(self, s: _XSlot) -> Tuple[int, int]:
        base = np.uint32((self.seed * 1_000_003 + s.trial_index * 4099 + s.epoch * 0x2545F491 + 77) & 0xFFFFFFFF)
        lo = int(mix32(base))
        hi = int(mix32(np.uint32(lo ^ 0x9E3779B9)))
        return lo, hi

    def _compile(self, eval_mode: bool, ids: Optional[List[int]] = None, tag: Optional[str] = None) -> Dict[str, Any]:
        """The program of the slots ``ids`` (default: all) for training steps (``eval_mode`` False) or for one validation
        chunk."""
        T, S, fp = self.capacity, self.S, self.fpad
        tag = tag or ("e" if eval_mode else "t")
        when = 2 if eval_mode else 1
        ids = sorted(self.slots) if ids is None else sorted(ids)
        out: Dict[str, Any] = {"ids": ids}
        members = [(slot, self.slots[slot]) for slot in ids]
        ids_ptr = self._upload(tag + "ids", np.array(ids, dtype=np.int32))
        pP, pG = self.P.data_ptr(), self.G.data_ptr()
        seeds = self.seeds.data_ptr()
        grp = L.gemm_group
        ops: List[XfOp] = []
        counter = [0]

        def name() -> str:
            counter[0] += 1
            return f"{tag}{counter[0]}"

        def op(kind: int, n: int, p0: int = 0, p1: int = 0, a: int = 0, b: int = 0, c: int = 0, d: int = 0, flags: int = 0,
               p2: int = 0, p3: int = 0) -> None:
            o = XfOp()
            o.kind, o.when, o.n, o.a, o.b, o.c, o.d, o.flags = kind, when, n, a, b, c, d, flags
            o.p0, o.p1, o.p2, o.p3 = p0 or None, p1 or None, p2 or None, p3 or None
            ops.append(o)

        info: Dict[int, Dict[str, Any]] = {}
        for slot, s in members:
            sp = s.spec
            B = self.eval_chunk if eval_mode else min(sp.opt.batch_size, self.n_train)
            steps = 1 if eval_mode else self.n_train // B
            p = 0.0 if eval_mode else sp.dropout
            base = s.buf.data_ptr()
            shared = sp.shared

            def Pf(key: str, l: Optional[int] = None, s=s, slot=slot) -> int:
                k = key if l is None else f"layers.{0 if s.spec.shared else l}.{key}"
                return pP + (slot * self.stride + s.layout.off[k]) * 4

            def Gf(key: str, l: Optional[int] = None, s=s, slot=slot, base=base) -> int:
                if l is None:
                    return pG + (slot * self.stride + s.layout.off[key]) * 4
                if s.spec.shared:       # this layer's own copy; summed into the slab after the backward pass
                    rel = s.layout.off[f"layers.0.{key}"] - s.layout.layer_lo[0]
                    return base + (s.arena.off["GL"] + l * s.layout.layer_span + rel) * 4
                return pG + (slot * self.stride + s.layout.off[f"layers.{l}.{key}"]) * 4

            info[slot] = dict(B=B, m=B * S, steps=steps, thresh=min(int(p * 4294967296.0), 0xFFFFFFFF) if p > 0 else 0,
                              keep=1.0 / (1.0 - p) if p > 0 else 1.0, P=Pf, G=Gf,
                              A=lambda key, s=s, base=base: base + s.arena.off[key] * 4, seed=seeds + slot * 8)

        def gemm(mode: int, recs: List[Tuple[int, tuple, int, int, int]], flags: int = 0, a_shared: int = 0,
                 rows_kind: int = 0) -> None:
            """One grouped GEMM launch: ``recs`` = (unused, record, out rows, out cols, weight).  The tile edge follows the
            launch's size like the MLP engine's: 128 when that still leaves thousands of tiles, 32 when 64-tiles would not
            fill the chip.  Weight gradients (mode 2) of small layers are a handful of tiles as long as the trial's token
            count: those groups split the contraction, every split writes its partial result to a scratch slab, and a sum
            kernel adds the slabs in split order into the gradient (fixed order: bitwise repeatable)."""
            if not recs:
                return
            count = lambda e: sum(((r + e - 1) // e) * ((c + e - 1) // e) for _, _, r, c, _ in recs)      # noqa: E731
            # bf16 products leave the tile bound by LDS reads and DMA instead of the matrix pipe: 128-tiles halve both per flop
            big = OPTIONS.xf_bf16_big_tiles if self.product_flag else OPTIONS.f32_small_tiles
            edge = 128 if count(128) >= big else 64
            if edge == 64 and count(64) < OPTIONS.xf_tiny_tiles:
                edge = 32
            if mode == 2:
                edge = min(edge, 64)        # weight gradients: few tiles, each as long as the token count
            groups = np.zeros(len(recs), dtype=L.GEMM_GROUP_DTYPE)
            shapes, sums, slab = [], [], 0
            for k, (_, rec, r, c, w) in enumerate(recs):
                rec = list(rec)
                ti, tj = (r + edge - 1) // edge, (c + edge - 1) // edge
                ks = 1
                if mode == 2:
                    C0, db, KK = rec[2], rec[4], rec[8]
                    n = r * c + (r if db else 0)
                    # the split count follows the layer's OWN shape (its 64-tile count), not the edge this launch happens
                    # to use: the edge depends on who else is in the pack, and the split count decides the order in which
                    # a gradient is summed -- a trial's numbers may not depend on its neighbours
                    t64 = ((r + 63) // 64) * ((c + 63) // 64)
                    ks = max(1, min(16, (WGRAD_TARGET_TILES + t64 - 1) // t64, (KK // 32) // 4))
                    if ks > 1 and n % 4 == 0 and (not db or db == C0 + r * c * 4):
                        sums.append((k, C0, slab, n, ks, rec[13]))
                        slab += n
                    else:
                        ks = 1
                    rec[17] = ks
                groups[k] = tuple(rec)
                shapes.append((k, ti, tj, ks))
            stride = 0
            if sums:
                stride = (slab + 63) // 64 * 64
                need = stride * max(x[4] for x in sums)
                if out.get("wg_scratch") is None or out["wg_scratch"].numel() < need:
                    out.setdefault("keep", []).append(out.get("wg_scratch"))       # earlier launches keep theirs
                    out["wg_scratch"] = torch.empty(max(need, 2 * (out["wg_scratch"].numel() if out.get("wg_scratch") is not None
                                                                   else 0)), device=self.device, dtype=torch.float32)
                base = out["wg_scratch"].data_ptr()
                for k, C0, off, n, ks, steps in sums:
                    groups[k]["C"] = base + off * 4
                    if groups[k]["aux_out"]:
                        groups[k]["aux_out"] = base + (off + int(groups[k]["I"]) * int(groups[k]["J"])) * 4
            tiles = spread_tiles(shapes)
            nm = name()
            op(K_GEMM, len(tiles), self._upload(nm + "g", groups), self._upload(nm + "t", tiles), a=mode, b=edge, c=a_shared,
               d=rows_kind, flags=flags | self.product_flag, p2=stride)
            if sums:
                tab = np.zeros(len(sums), dtype=SUM_DTYPE)
                for j, (k, C0, off, n, ks, steps) in enumerate(sums):
                    tab[j] = (base + off * 4, C0, stride, n, ks, steps, 0)
                op(K_SUM, len(sums), self._upload(nm + "s", tab), self._upload(nm + "si", np.arange(len(sums), dtype=np.int32)),
                   a=max(x[3] for x in sums))

        def table(dtype, fill: Callable[[int, _XSlot, Dict[str, Any]], Optional[tuple]]) -> Tuple[int, int]:
            arr = np.zeros(T, dtype=dtype)
            n = 0
            for slot, s in members:
                rec = fill(slot, s, info[slot])
                if rec is not None:
                    arr[slot] = rec
                    n += 1
            return (self._upload(name(), arr), len(ids) if n else 0)       # kernels walk all ids; empty records return at once

        max_b = max(i["B"] for i in info.values())
        max_m = max_b * S
        max_d = max(s.spec.d for _, s in members)
        Lmax = max(s.spec.layers for _, s in members)

        # ---- batch, input projection, positions
        if not eval_mode:
            gat = np.zeros(T, dtype=L.GATHER_DTYPE)       # keys and step counts are per epoch: refreshed by _epoch_tables
            out["gather"] = name()
            op(K_GATHER, len(ids), self._upload(out["gather"], gat), ids_ptr, a=max_b, b=self.row_len,
               c=half_bits(self.n_train), d=self.n_train, p2=self.Xtr.data_ptr(), p3=self.ytr.data_ptr())
        recs = []
        for slot, s in members:
            i, d = info[slot], s.spec.d
            a_in = self.Xva.data_ptr() if eval_mode else i["A"]("X0")
            recs.append((slot, grp(a_in, i["P"]("input_projection.weight"), i["A"]("H0"), i["P"]("input_projection.bias"), 0,
                                   i["m"], d, fp, fp, fp, d, 0, i["steps"]), i["m"], d, 1))
        gemm(0, recs, a_shared=1 if eval_mode else 0)
        ptr, n = table(PE_DTYPE, lambda slot, s, i: (i["A"]("H0"), i["A"]("pe"), i["A"]("dH"), i["seed"], i["m"], S, s.spec.d,
                                                      i["steps"], i["keep"], i["thresh"], PE_SITE, 0))
        op(K_PE_FWD, n, ptr, ids_ptr, a=max_m, b=max_d)
        pe_table = ptr

        backward: List[Callable[[], None]] = []          # per layer, appended in forward order, run reversed

        for l in range(Lmax):
            act = [(slot, s) for slot, s in members if l < s.spec.layers]
            h_in = {slot: (info[slot]["A"]("H0") if l == 0 else info[slot]["A"](f"H2_{l - 1}")) for slot, _ in act}
            site = 8 * l

            def live(fn):          # a table over the slots that have this layer (and satisfy fn's own condition)
                return lambda slot, s, i, l=l: fn(slot, s, i) if l < s.spec.layers else None

            # fused projection
            gemm(0, [(slot, grp(h_in[slot], info[slot]["P"]("attention.qkv_proj.weight", l), info[slot]["A"](f"QKV{l}"),
                                info[slot]["P"]("attention.qkv_proj.bias", l), 0, info[slot]["m"], s.spec.nq, s.spec.d, s.spec.d,
                                s.spec.d, s.spec.nq, 0, info[slot]["steps"]), info[slot]["m"], s.spec.nq, 1) for slot, s in act])

            # local value mixing (relu attention): VM = v + conv(v)
            def mix_rec(slot, s, i, l=l):
                sp = s.spec
                if not sp.ksize:
                    return None
                A, voff = i["A"], 2 * sp.heads * sp.dk * 4
                return (A(f"QKV{l}") + voff, i["P"]("attention.local.weight", l), i["P"]("attention.local.bias", l), A(f"VM{l}"),
                        A("dVM"), 0, A("dQKV") + voff, A("cpartA"), i["G"]("attention.local.weight", l),
                        i["G"]("attention.local.bias", l), i["m"], S, sp.d, sp.ksize, sp.nq, sp.d, sp.d, sp.nq, 1, i["steps"])
            mix_ptr, mix_n = table(CONV_DTYPE, live(mix_rec))
            op(K_CONV_FWD, mix_n, mix_ptr, ids_ptr, a=max_m, b=max_d)

            # attention: fused kernels for heads up to 64 wide ...
            def attn_rec(slot, s, i, l=l, site=site):
                sp = s.spec
                if not sp.fused:
                    return None
                A = i["A"]
                q, dq = A(f"QKV{l}"), A("dQKV")
                hk = sp.heads * sp.dk * 4
                v, ldv = (A(f"VM{l}"), sp.d) if sp.ksize else (q + 2 * hk, sp.nq)
                dvp = A("dVM") if sp.ksize else dq + 2 * hk
                return attn_slot(q, q + hk, v, A(f"O{l}"), A(f"astat{l}"), A("dO"), dq, dq + hk, dvp, i["seed"], i["B"], sp.heads,
                                 sp.dk, sp.dv, sp.nq, ldv, sp.d, sp.kind, i["steps"], site, 1.0 / math.sqrt(sp.dk), i["keep"],
                                 i["thresh"])
            at_ptr, at_n = table(ATTN_DTYPE, live(attn_rec))
            fused = [(slot, s) for slot, s in act if s.spec.fused]
            if fused:
                max_bh = max(info[slot]["B"] * s.spec.heads for slot, s in fused)
                wide = max(2, max(max(s.spec.dk, s.spec.dv) for _, s in fused))
                op(K_ATTN_FWD, at_n, at_ptr, ids_ptr, a=max_bh, b=S, c=wide, d=max(s.spec.heads for _, s in fused))
            # ... explicit score matrices for wider ones
            wide_slots = [(slot, s) for slot, s in act if not s.spec.fused]
            comp = self._score_products(wide_slots, info, l, name, op, ids_ptr, site, len(ids)) if wide_slots else None
            if comp is not None:
                comp["forward"]()

            # output projection, residual + LayerNorm 1
            gemm(0, [(slot, grp(info[slot]["A"](f"O{l}"), info[slot]["P"]("attention.out_proj.weight", l), info[slot]["A"]("T"),
                                info[slot]["P"]("attention.out_proj.bias", l), 0, info[slot]["m"], s.spec.d, s.spec.d, s.spec.d,
                                s.spec.d, s.spec.d, 0, info[slot]["steps"]), info[slot]["m"], s.spec.d, 1) for slot, s in act])

            def ln_rec(which: int):
                def fill(slot, s, i, l=l, site=site, which=which):
                    sp, A = s.spec, i["A"]
                    rate = 0.0 if eval_mode else sp.drop_path(l)
                    sd_t = min(int(rate * 4294967296.0), 0xFFFFFFFF) if rate > 0 else 0
                    n1 = f"norm{which}"
                    if which == 1:
                        x, res, y, z, st, dx, part = A("T"), h_in[slot], A(f"H1_{l}"), A(f"Z1_{l}"), A(f"st1_{l}"), A("dB1"), A(f"part1_{l}")
                    else:
                        x, res, y, z, st, dx, part = A("T"), A(f"H1_{l}"), A(f"H2_{l}"), A(f"Z2_{l}"), A(f"st2_{l}"), A("dB2"), A(f"part2_{l}")
                    return (x, res, i["P"](n1 + ".weight", l), i["P"](n1 + ".bias", l), y, z, st, A("dH"), dx, A("dR"), part,
                            i["G"](n1 + ".weight", l), i["G"](n1 + ".bias", l), i["seed"], i["m"], sp.d, i["steps"],
                            site + (1 if which == 1 else 3), i["keep"], 1e-5, i["thresh"], S, sd_t,
                            1.0 / (1.0 - rate) if rate > 0 else 1.0, site + (4 if which == 1 else 5), 0)
                return fill
            ln1_ptr, ln_n = table(LN_DTYPE, live(ln_rec(1)))
            op(K_LN_FWD, ln_n, ln1_ptr, ids_ptr, a=max_m, b=max_d)

            # feed-forward: dense, or depthwise convolution + two pointwise layers
            def ffc_rec(slot, s, i, l=l):
                sp, A = s.spec, i["A"]
                if not sp.depthwise:
                    return None
                return (A(f"H1_{l}"), i["P"]("feedforward.depthwise.weight", l), i["P"]("feedforward.depthwise.bias", l),
                        A(f"C1_{l}"), A("dC1"), A("dR"), A("dH"), A("cpartF"), i["G"]("feedforward.depthwise.weight", l),
                        i["G"]("feedforward.depthwise.bias", l), i["m"], S, sp.d, 3, sp.d, sp.d, sp.d, sp.d, 0, i["steps"])
            ffc_ptr, ffc_n = table(CONV_DTYPE, live(ffc_rec))
            op(K_CONV_FWD, ffc_n, ffc_ptr, ids_ptr, a=max_m, b=max_d)

            def ffkeys(sp: XfSpec, l=l) -> Tuple[str, str, str]:
                return (("feedforward.pointwise", "feedforward.project", f"C1_{l}") if sp.depthwise
                        else ("feedforward.0", "feedforward.2", f"H1_{l}"))
            recs1, recs2 = [], []
            for slot, s in act:
                sp, i = s.spec, info[slot]
                k1, k2, src = ffkeys(sp)
                recs1.append((slot, grp(i["A"](src), i["P"](k1 + ".weight", l), i["A"](f"F1_{l}"), i["P"](k1 + ".bias", l), 0,
                                        i["m"], sp.dff, sp.d, sp.d, sp.d, sp.dff, 0, i["steps"], gflags=L.GROUP_ACT),
                              i["m"], sp.dff, 1))
                recs2.append((slot, grp(i["A"](f"F1_{l}"), i["P"](k2 + ".weight", l), i["A"]("T"), i["P"](k2 + ".bias", l), 0,
                                        i["m"], sp.d, sp.dff, sp.dff, sp.dff, sp.d, 0, i["steps"]), i["m"], sp.d, 2))
            gemm(0, recs1)
            gemm(0, recs2)
            ln2_ptr, _ = table(LN_DTYPE, live(ln_rec(2)))
            op(K_LN_FWD, ln_n, ln2_ptr, ids_ptr, a=max_m, b=max_d)

            if eval_mode:
                continue

            def layer_backward(l=l, act=act, h_in=h_in, ln1_ptr=ln1_ptr, ln2_ptr=ln2_ptr, ln_n=ln_n, ffc_ptr=ffc_ptr, ffc_n=ffc_n,
                               mix_ptr=mix_ptr, mix_n=mix_n, at_ptr=at_ptr, at_n=at_n, fused=fused, comp=comp, ffkeys=ffkeys):
                op(K_LN_BWD, ln_n, ln2_ptr, ids_ptr, a=max_m, b=max_d)
                op(K_LN_REDUCE, ln_n, ln2_ptr, ids_ptr)
                wg: List[Tuple[int, tuple, int, int, int]] = []
                d2, d1, dpw = [], [], []
                for slot, s in act:
                    sp, i = s.spec, info[slot]
                    A, P, G, m, st, d, dff = i["A"], i["P"], i["G"], i["m"], i["steps"], sp.d, sp.dff
                    k1, k2, src = ffkeys(sp)
                    kw_ = (m + 63) // 64
                    wg.append((0, grp(A("dB2"), A(f"F1_{l}"), G(k2 + ".weight", l), 0, G(k2 + ".bias", l), d, dff, m, d, dff, dff,
                                      0, st), d, dff, kw_))
                    d2.append((0, grp(A("dB2"), P(k2 + ".weight", l), A("dF1"), A(f"F1_{l}"), 0, m, dff, d, d, dff, dff, dff, st,
                                      gflags=L.GROUP_ACT), m, dff, 1))
                    wg.append((0, grp(A("dF1"), A(src), G(k1 + ".weight", l), 0, G(k1 + ".bias", l), dff, d, m, dff, d, d, 0, st),
                               dff, d, kw_))
                    if sp.depthwise:
                        dpw.append((0, grp(A("dF1"), P(k1 + ".weight", l), A("dC1"), 0, 0, m, d, dff, dff, d, d, 0, st), m, d, 4))
                    else:
                        d1.append((0, grp(A("dF1"), P(k1 + ".weight", l), A("dH"), A("dR"), 0, m, d, dff, dff, d, d, d, st,
                                          gflags=L.GROUP_ADD), m, d, 4))
                gemm(1, d2)
                gemm(1, d1 + dpw)
                op(K_CONV_WGRAD, ffc_n, ffc_ptr, ids_ptr, a=max_m, b=max_d)
                op(K_CONV_BWD, ffc_n, ffc_ptr, ids_ptr, a=max_m, b=max_d)
                op(K_LN_BWD, ln_n, ln1_ptr, ids_ptr, a=max_m, b=max_d)
                op(K_LN_REDUCE, ln_n, ln1_ptr, ids_ptr)
                dout = []
                for slot, s in act:
                    sp, i = s.spec, info[slot]
                    A, P, G, m, st, d = i["A"], i["P"], i["G"], i["m"], i["steps"], sp.d
                    wg.append((0, grp(A("dB1"), A(f"O{l}"), G("attention.out_proj.weight", l), 0, G("attention.out_proj.bias", l),
                                      d, d, m, d, d, d, 0, st), d, d, (m + 63) // 64))
                    dout.append((0, grp(A("dB1"), P("attention.out_proj.weight", l), A("dO"), 0, 0, m, d, d, d, d, d, 0, st), m, d, 1))
                gemm(1, dout)
                if fused:
                    op(K_ATTN_BWD, at_n, at_ptr, ids_ptr, a=max(info[slot]["B"] * s.spec.heads for slot, s in fused), b=S,
                       c=max(2, max(max(s.spec.dk, s.spec.dv) for _, s in fused)))
                if comp is not None:
                    comp["backward"]()
                op(K_CONV_WGRAD, mix_n, mix_ptr, ids_ptr, a=max_m, b=max_d)
                op(K_CONV_BWD, mix_n, mix_ptr, ids_ptr, a=max_m, b=max_d)
                dq = []
                for slot, s in act:
                    sp, i = s.spec, info[slot]
                    A, P, G, m, st, d, nq = i["A"], i["P"], i["G"], i["m"], i["steps"], sp.d, sp.nq
                    wg.append((0, grp(A("dQKV"), h_in[slot], G("attention.qkv_proj.weight", l), 0, G("attention.qkv_proj.bias", l),
                                      nq, d, m, nq, d, d, 0, st), nq, d, (m + 63) // 64))
                    dq.append((0, grp(A("dQKV"), P("attention.qkv_proj.weight", l), A("dH"), A("dR"), 0, m, d, nq, nq, d, d, d, st,
                                      gflags=L.GROUP_ADD), m, d, 3))
                # the layer's weight gradients in one launch: their operands are complete and not yet overwritten
                gemm(2, wg, flags=1)
                gemm(1, dq)
            backward.append(layer_backward)

        # ---- head on the last token of every sample
        def last(slot: int, s: _XSlot, key: str) -> int:
            return info[slot]["A"](key) + (S - 1) * s.spec.d * 4

        dims = (None,) + HEAD_DIMS
        for j in range(1, 5):
            recs = []
            for slot, s in members:
                i, sp = info[slot], s.spec
                k_in = sp.d if j == 1 else dims[j - 1]
                a_in, lda = (last(slot, s, f"H2_{sp.layers - 1}"), S * sp.d) if j == 1 else (i["A"](f"A{j - 1}"), k_in)
                recs.append((slot, grp(a_in, i["P"](f"fc{j}.weight"), i["A"](f"A{j}"), i["P"](f"fc{j}.bias"), 0, i["B"], dims[j],
                                       k_in, lda, k_in, dims[j], 0, i["steps"], gflags=L.GROUP_ACT), i["B"], dims[j], 1))
            gemm(0, recs, rows_kind=1)
        head = np.zeros(T, dtype=L.HEAD_DTYPE)
        for slot, s in members:
            i = info[slot]
            head[slot] = (i["A"]("A4"), i["A"]("dA4"), i["P"]("fc5.weight"), i["P"]("fc5.bias"), i["G"]("fc5.weight"),
                          i["G"]("fc5.bias"), self.yva.data_ptr() if eval_mode else i["A"]("y"),
                          self.loss_acc.data_ptr() + slot * 4, self.metrics.data_ptr() + slot * 24, i["B"], HEAD_DIMS[-1],
                          HEAD_DIMS[-1], LOSS_CODES[s.spec.opt.loss_function], i["steps"], 0)
        head_ptr = self._upload(name(), head)
        # the head kernels' partial records are laid out by (slot, row chunk of THIS launch): programs that run next to
        # each other on different streams must not share them
        out["scratch"] = self._tables.setdefault(tag + "scratch", torch.zeros_like(self.head_scratch))
        scratch_ptr = out["scratch"].data_ptr()
        if eval_mode:
            op(K_HEAD_EVAL, len(ids), head_ptr, ids_ptr, a=self.eval_chunk, b=HEAD_DIMS[-1], p2=scratch_ptr)
            return self._finish(ops, out)

        op(K_HEAD_LOSS, len(ids), head_ptr, ids_ptr, a=max_b, b=HEAD_DIMS[-1], p2=scratch_ptr)
        ptr, n = table(FILL_DTYPE, lambda slot, s, i: (i["A"]("dH"), i["m"] * s.spec.d, i["steps"], 0))
        op(K_FILL, n, ptr, ids_ptr, a=max_m * max_d)
        wg = []
        for j in range(4, 0, -1):
            dg = []
            for slot, s in members:
                i, sp = info[slot], s.spec
                A, P, G, B, st = i["A"], i["P"], i["G"], i["B"], i["steps"]
                k_in = sp.d if j == 1 else dims[j - 1]
                a_in, lda = (last(slot, s, f"H2_{sp.layers - 1}"), S * sp.d) if j == 1 else (A(f"A{j - 1}"), k_in)
                wg.append((0, grp(A(f"dA{j}"), a_in, G(f"fc{j}.weight"), 0, G(f"fc{j}.bias"), dims[j], k_in, B, dims[j], lda, k_in,
                                  0, st), dims[j], k_in, 1))
                if j > 1:
                    dg.append((0, grp(A(f"dA{j}"), P(f"fc{j}.weight"), A(f"dA{j - 1}"), A(f"A{j - 1}"), 0, B, k_in, dims[j], dims[j],
                                      k_in, k_in, k_in, st, gflags=L.GROUP_ACT), B, k_in, 1))
                else:
                    dg.append((0, grp(A("dA1"), P("fc1.weight"), last(slot, s, "dH"), 0, 0, B, k_in, dims[1], dims[1], k_in,
                                      S * sp.d, 0, st), B, k_in, 1))
            gemm(1, dg)
        gemm(2, wg, flags=1)
        for fn in reversed(backward):
            fn()
        op(K_PE_BWD, len(ids), pe_table, ids_ptr, a=max_m, b=max_d)
        gemm(2, [(0, grp(info[slot]["A"]("dH"), info[slot]["A"]("X0"), info[slot]["G"]("input_projection.weight"), 0,
                         info[slot]["G"]("input_projection.bias"), s.spec.d, fp, info[slot]["m"], s.spec.d, fp, fp, 0,
                         info[slot]["steps"]), s.spec.d, fp, (info[slot]["m"] + 63) // 64) for slot, s in members], flags=1)
        # gradients of shared layers: the layers' copies added in layer order into the slab
        def sum_rec(slot, s, i):
            if not s.spec.shared:
                return None
            return (i["A"]("GL"), pG + (slot * self.stride + s.layout.layer_lo[0]) * 4, s.layout.layer_span, s.layout.layer_span,
                    s.spec.layers, i["steps"], 0)
        ptr, n = table(SUM_DTYPE, sum_rec)
        op(K_SUM, n, ptr, ids_ptr, a=max((s.layout.layer_span for _, s in members if s.spec.shared), default=0))
        q = out["optim"] = XfOptim()
        q.P, q.G, q.M, q.V = self.P.data_ptr(), self.G.data_ptr(), self.M.data_ptr(), self.V.data_ptr()
        q.partials, q.slot_stride = self.partials.data_ptr(), self.stride
        q.max_params = max(s.layout.total for _, s in members)
        q.max_blocks = self.max_blocks
        q.use_clip = int(any(s.spec.opt.gradient_clipping > 0 for _, s in members))
        out["opt"] = name()
        op(K_OPTIM, len(ids), self._upload(out["opt"], np.zeros(T, dtype=L.OPT_DTYPE)), ids_ptr, p2=C.addressof(q))
        return self._finish(ops, out)

    def _score_products(self, wide_slots, info, l, name, op, ids_ptr, site, n_ids) -> Dict[str, Callable[[], None]]:
        """Attention of the slots whose heads are too wide for the fused kernels, as grouped GEMMs per (sample, head):
        scores = q k^T, a row kernel (scale, softmax or relu-normalise, dropout), o = p v; and backwards dP = dO v^T,
        dV = p^T dO, the row kernel's gradient, dQ = dS k, dK = dS^T q.  All exact f32."""
        S = self.S
        T = self.capacity
        fl = lambda x: x                                    # noqa: E731

        def groups_of(kind: str) -> Tuple[np.ndarray, int, int]:
            """All (sample, head) groups of one product over the wide slots; returns (records, out rows, out cols)."""
            parts = []
            r = c = 0
            for slot, s in wide_slots:
                sp, i = s.spec, info[slot]
                A, B, H, nq, d, dk, dv = i["A"], i["B"], sp.heads, sp.nq, sp.d, sp.dk, sp.dv
                bh = np.arange(B * H, dtype=np.int64)
                b, h = bh // H, bh % H
                q0, dq0 = A(f"QKV{l}"), A("dQKV")
                qp = q0 + (b * S * nq + h * dk) * 4
                kp = qp + H * dk * 4
                dqp = dq0 + (b * S * nq + h * dk) * 4
                dkp = dqp + H * dk * 4
                if sp.ksize:
                    vp, ldv = A(f"VM{l}") + (b * S * d + h * dv) * 4, d
                    dvp = A("dVM") + (b * S * d + h * dv) * 4
                else:
                    vp, ldv = q0 + (b * S * nq + 2 * H * dk + h * dv) * 4, nq
                    dvp = dq0 + (b * S * nq + 2 * H * dk + h * dv) * 4
                op_ = A(f"O{l}") + (b * S * d + h * dv) * 4
                dop = A("dO") + (b * S * d + h * dv) * 4
                pn = A(f"Pn{l}") + bh * S * S * 4
                pd = (A(f"Pd{l}") + bh * S * S * 4) if f"Pd{l}" in s.arena.off and i["thresh"] else pn
                dp = A("dP") + bh * S * S * 4
                g = np.zeros(B * H, dtype=L.GEMM_GROUP_DTYPE)
                g["active_steps"] = i["steps"]
                if kind == "qk":        # mode 0: scores[S x S] = q [S x dk] . k [S x dk]^T
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = qp, kp, pn, S, S, dk, nq, nq, S
                    r, c = S, S
                elif kind == "pv":      # mode 1: o [S x dv] = p [S x S] . v [S x dv]
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = pd, vp, op_, S, dv, S, S, ldv, d
                    r, c = S, max(c, dv)
                elif kind == "dp":      # mode 0: dP [S x S] = dO [S x dv] . v [S x dv]^T
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = dop, vp, dp, S, S, dv, d, ldv, S
                    r, c = S, S
                elif kind == "dv":      # mode 2: dV [S x dv] = p [S x S]^T . dO [S x dv]
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = pd, dop, dvp, S, dv, S, S, d, ldv
                    r, c = S, max(c, dv)
                elif kind == "dq":      # mode 1: dQ [S x dk] = dS [S x S] . k [S x dk]
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = dp, kp, dqp, S, dk, S, S, nq, nq
                    r, c = S, max(c, dk)
                else:                   # "dk", mode 2: dK [S x dk] = dS [S x S]^T . q [S x dk]
                    g["A"], g["B"], g["C"], g["I"], g["J"], g["KK"], g["lda"], g["ldb"], g["ldc"] = dp, qp, dkp, S, dk, S, S, nq, nq
                    r, c = S, max(c, dk)
                parts.append(g)
            return np.concatenate(parts), r, c

        def launch(kind: str, mode: int) -> None:
            g, r, c = groups_of(kind)
            edge = 32
            tiles = _uniform_tiles(len(g), (r + edge - 1) // edge, (c + edge - 1) // edge)
            nm = name()
            op(K_GEMM, len(tiles), self._upload(nm + "g", g), self._upload(nm + "t", tiles), a=mode, b=edge, c=0, d=2,
               flags=self.product_flag)

        rows = np.zeros(T, dtype=ROW_DTYPE)
        for slot, s in wide_slots:
            sp, i = s.spec, info[slot]
            A = i["A"]
            pn = A(f"Pn{l}")
            pd = A(f"Pd{l}") if f"Pd{l}" in s.arena.off and i["thresh"] else pn
            rows[slot] = (pn, pd, A(f"astat{l}"), A("dP"), i["seed"], i["B"] * sp.heads, S, sp.kind, i["steps"], sp.heads, site,
                          1.0 / math.sqrt(sp.dk), i["keep"], i["thresh"], 0)
        row_ptr = self._upload(name(), rows)
        max_bh = max(info[slot]["B"] * s.spec.heads for slot, s in wide_slots)
        max_h = max(s.spec.heads for _, s in wide_slots)
        n = n_ids                        # the kernels walk every slot of the program; the others' records are empty

        def forward() -> None:
            launch("qk", 0)
            op(K_ROW_FWD, n, row_ptr, ids_ptr, a=max_bh, b=S, d=max_h)
            launch("pv", 1)

        def backward() -> None:
            launch("dp", 0)
            launch("dv", 2)
            op(K_ROW_BWD, n, row_ptr, ids_ptr, a=max_bh, b=S, d=max_h)
            launch("dq", 1)
            launch("dk", 2)

        return {"forward": forward, "backward": backward}

    def _finish(self, ops: List[XfOp], out: Dict[str, Any]) -> Dict[str, Any]:
        out["prog"], out["n"] = (XfOp * len(ops))(*ops), len(ops)
        return out

    def _refresh(self) -> None:
        """Programs of the current pack: one validation program over all slots, and one training program per CHAIN.  A
        chain holds the slots of one batch size -- they take the same number of steps per epoch, so no launch of a chain
        ever carries finished slots -- and runs on a HIP stream of its own, next to the others (at most ``MAX_CHAINS``;
        with more batch sizes than that the rarest ones share a chain)."""
        if not (self._dirty or not self._chains):
            return
        by_batch: Dict[int, List[int]] = {}
        for slot, s in self.slots.items():
            by_batch.setdefault(min(s.spec.opt.batch_size, self.n_train), []).append(slot)
        groups = sorted(by_batch.values(), key=lambda g: (-len(g), g[0]))
        while len(groups) > MAX_CHAINS:
            groups[-2] = groups[-2] + groups[-1]
            groups.pop()
        self._arena_turn ^= 1
        capacity = self._arenas[self._arena_turn].capacity if self._arenas[self._arena_turn] is not None else (8 << 20)
        while True:
            arena = self._arenas[self._arena_turn]
            if arena is None or arena.capacity < capacity:
                arena = self._arenas[self._arena_turn] = _TableArena(self.device, capacity)
            arena.reset()
            self._arena, self._arena_open = arena, True
            try:
                self._chains = [self._compile(False, g, f"t{c}_") for c, g in enumerate(groups)]
                self._eval = self._compile(True)
                break
            except _ArenaFull:
                capacity *= 2
            finally:
                self._arena_open = False
        arena.flush()
        while len(self._streams) < len(self._chains) - 1:
            self._streams.append(torch.cuda.Stream(self.device))
        self._dirty = False

    def _epoch_tables(self) -> int:
        T = self.capacity
        gat = np.zeros(T, dtype=L.GATHER_DTYPE)
        opt = np.zeros(T, dtype=L.OPT_DTYPE)
        seeds = np.zeros((T, 2), dtype=np.uint32)
        max_steps = 0
        for slot, s in self.slots.items():
            sp = s.spec.opt
            b = min(sp.batch_size, self.n_train)
            steps = self.n_train // b
            max_steps = max(max_steps, steps)
            base = s.buf.data_ptr()
            gat[slot] = (base + s.arena.off["X0"] * 4, base + s.arena.off["y"] * 4, round_keys(self.seed, s.trial_index, s.epoch),
                         b, steps)
            opt[slot] = (sp.learning_rate, sp.beta1, sp.beta2, sp.eps, sp.weight_decay, sp.momentum, sp.gradient_clipping,
                         OPT_CODES[sp.optimizer], sp.warmup_steps, sp.total_steps, s.layout.total, steps, s.step, 0, 0, 0)
            seeds[slot] = self._slot_seed(s)
        for ch in self._chains:          # the per-epoch tables are indexed by slot: every chain gets the same ones
            self._upload(ch["gather"], gat)
            self._upload(ch["opt"], opt)
        self.seeds.copy_(torch.from_numpy(seeds.view(np.int32)))
        return max_steps

    # ------------------------------------------------------------------ training / evaluation
    def train_epoch(self):
        if not self.slots:
            return
        self._refresh()
        self._epoch_tables()
        self.loss_acc.zero_()
        main = torch.cuda.current_stream(self.device)
        stream = main.cuda_stream
        lanes = [main] + self._streams[:len(self._chains) - 1]
        for st in lanes[1:]:
            st.wait_stream(main)             # tables, seeds and zeroed accumulators were issued on the main stream
        steps = [max(self.n_train // min(self.slots[k].spec.opt.batch_size, self.n_train) for k in ch["ids"])
                 for ch in self._chains]
        for step in range(max(steps)):       # interleaved step by step: no chain gets a head start
            for ch, st, n_steps in zip(self._chains, lanes, steps):
                if step < n_steps:
                    L.check(self.lib.mt_xf_run(C.cast(ch["prog"], c_void_p), ch["n"], step, 1, 1, 0, 0, self.S, st.cuda_stream),
                            "mt_xf_run")
        for st in lanes[1:]:
            main.wait_stream(st)
        L.check(self.lib.mt_stamp_launch(self._stamps[1:].data_ptr(), stream), "mt_stamp_launch")
        self._round_steps = {}
        for slot, s in self.slots.items():
            k = self.n_train // min(s.spec.opt.batch_size, self.n_train)
            self._round_steps[slot] = k
            s.step += k
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
        ev = self._eval
        for c0 in range(0, self.n_val, self.eval_chunk):
            nb = min(self.eval_chunk, self.n_val - c0)
            L.check(self.lib.mt_xf_run(C.cast(ev["prog"], c_void_p), ev["n"], 0, 1, 0, nb, c0, self.S, stream),
                    "mt_xf_run (evaluation)")

    def device_table(self) -> torch.Tensor:
        T = self.capacity
        meta = np.zeros(T, dtype=L.TABLE_META_DTYPE)
        for slot, s in self.slots.items():
            meta[slot] = (float(s.trial_index), float(s.epoch), self._round_steps.get(slot, 1), 1)
        if self.slots:
            self._enqueue_eval()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # through a pinned image and a posted copy: a blocking upload would hold the host until the whole round has run,
        # and the host has the next trials to prepare meanwhile (the image is rewritten a round later, after the table
        # has been read back)
        raw = meta.view(np.uint8).reshape(-1)
        if self._meta_host is None or self._meta_host.shape[1] != raw.size:
            self._meta_host = torch.empty(2, raw.size, dtype=torch.uint8).pin_memory()
            self._meta_copied = [None, None]
            self._tables["table_meta"] = torch.empty(raw.size, dtype=torch.uint8, device=self.device)
        k = self._meta_turn = self._meta_turn ^ 1
        if self._meta_copied[k] is not None:
            self._meta_copied[k].synchronize()      # rounds may be in flight (asynchronous exchange): two images, each reused
        self._meta_host[k].numpy()[:] = raw         # only once its last copy has been made
        self._tables["table_meta"].copy_(self._meta_host[k], non_blocking=True)
        self._meta_copied[k] = torch.cuda.Event()
        self._meta_copied[k].record()
        L.check(self.lib.mt_metric_table_launch(self._tables["table_meta"].data_ptr(), self.loss_acc.data_ptr(),
                                                self.metrics.data_ptr(), T, max(self.n_val, 1), len(self.slots),
                                                self._stamps.data_ptr(), self._table.data_ptr(), stream),
                "mt_metric_table_launch")
        self._loss_pending = False
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
