"""Dense layers on hand-written MFMA kernels -- forward, input gradient and weight gradient.

``linear(x, weight, bias)`` is ``torch.nn.functional.linear`` on the CPU (the oracle the GPU tests compare with).  On a
GPU no call reaches a BLAS-library GEMM:

* bf16 autocast: ``DenseBF16`` -- all three GEMMs on the 128x128 bf16 tile kernel of ``csrc/gemm.hip``;
* f32, small layers (at most ``SMALL_MAX_ROWS`` rows and ``in x out <= SMALL_MAX_WEIGHTS`` -- every projection of the
  sequence-regression Transformers at batch 32 x 50 tokens): ``SmallLinear`` -- forward on the exact-f32 64-tile kernel,
  and the backward's three products (input gradient, weight gradient, bias column sum) in ONE launch of
  ``mt_small_linear_bwd_tiled_launch`` (``csrc/small_linear.hip``: sixteen waves per 32x32 tile adding their
  accumulators through LDS in a fixed order; no workspace and no traffic between workgroups).  Measured per backward
  inside a replayed graph (three library launches / this one, us): 800x64x256 61.5 / 6.4, 1600x10x32 20.5 / 6.6,
  1600x64x256 29.9 / 6.9, 1600x128x384 27.5 / 13.0, 3200x64x256 43.7 / 18.3; past the bound larger LDS-staged tiles
  catch up (4800x256x768 77.1 / 74.4) and the layer goes to
* f32, everything else: ``DenseF32`` on the exact-f32 tile kernel of ``csrc/gemm_f32.hip``; the weight + bias gradient
  ``dW = dY^T X`` -- a long skinny contraction over the token rows -- is split over enough workgroups to fill the chip
  and the partial slabs are added in a fixed order by one small launch.

The tile kernels read rows in 16-byte pieces, so a dimension that is not a multiple of 4 (f32) or 8 (bf16) -- 81 input
features, a scalar output -- is padded with zero columns / zero rows up to the next multiple before the launch and the
result is sliced back; autograd carries the slice through, so gradients of the padding are dropped.  Operands these
kernels cannot take at all (f16 or f64 on a GPU) raise ``TypeError`` instead of taking a slower road silently.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from ..options import OPTIONS

SMALL_MAX_ROWS = 32768
SMALL_MAX_WEIGHTS = 65536     # in x out, e.g. 128 x 512
SMALL_MAX_MACS = 1 << 29      # rows x in x out; measured wins up to here: 6144x128x512 66.5 / 51.7 us, 12288x64x192
                              # 109.4 / 58.3, 24576x64x128 135.8 / 110.4 (library / fused)


def _shadow(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bf16 copy of a parameter kept current by the flat optimiser (``FlatParams(bf16_shadow=True)``), if there is one."""
    return getattr(t, "_mt_bf16", None) if t is not None else None


def _bf16_autocast() -> bool:
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16


def _small(rows: int, in_f: int, out_f: int, training: bool) -> bool:
    if in_f * out_f > SMALL_MAX_WEIGHTS or rows <= 0:
        return False
    return rows <= SMALL_MAX_ROWS and rows * in_f * out_f <= SMALL_MAX_MACS if training else rows <= 4 * SMALL_MAX_ROWS


def _pad_operands(x, weight, bias, mult: int):
    """Zero columns on x and weight up to a multiple of ``mult`` input features, zero rows on weight (and zeros on the
    bias) up to a multiple of ``mult`` output features."""
    out_f, in_f = weight.shape
    pi, po = (-in_f) % mult, (-out_f) % mult
    if pi:
        x, weight = F.pad(x, (0, pi)), F.pad(weight, (0, pi))
    if po:
        weight = F.pad(weight, (0, 0, 0, po))
        bias = F.pad(bias, (0, po)) if bias is not None else None
    return x, weight, bias


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ weight.T + bias``; on a GPU always on this package's own MFMA kernels (see the module text)."""
    if not x.is_cuda or x.numel() == 0:
        return F.linear(x, weight, bias)
    out_f, in_f = weight.shape
    if torch.is_autocast_enabled():
        if not _bf16_autocast() or weight.dtype != torch.float32:
            raise TypeError("mitune.ops.linear under autocast takes f32 master weights and bf16 compute only, got "
                            f"autocast {torch.get_autocast_dtype('cuda')} / weight {weight.dtype}")
        if out_f == 1:
            # a scalar head: a row-wise dot product in f32 (one multiply + one row reduction), not a GEMV
            with torch.autocast("cuda", enabled=False):
                y = (x.float() * weight.float().reshape(-1)).sum(-1, keepdim=True)
                return y if bias is None else y + bias.float()
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        if out_f % 8 == 0 and in_f % 8 == 0:
            return DenseBF16.apply(xb, weight, bias, _shadow(weight))
        xp, wp, bp = _pad_operands(xb, weight, bias, 8)
        y = DenseBF16.apply(xp, wp, bp, None)
        return y if wp.shape[0] == out_f else y[..., :out_f]
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        raise TypeError(f"mitune.ops.linear on a GPU takes f32 operands (or bf16 autocast), got x {x.dtype} / weight "
                        f"{weight.dtype}" + (f" / bias {bias.dtype}" if bias is not None else ""))
    rows = x.numel() // max(in_f, 1)
    training = torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)
    fn = SmallLinear if _small(rows, in_f, out_f, training) else DenseF32
    if out_f % 4 == 0 and in_f % 4 == 0:
        return fn.apply(x, weight, bias)
    xp, wp, bp = _pad_operands(x, weight, bias, 4)
    y = fn.apply(xp, wp, bp)
    return y if wp.shape[0] == out_f else y[..., :out_f]


def _wgrad_splits(M: int, tiles: int, rows_per_split: int) -> int:
    return max(1, min(127, (M + rows_per_split - 1) // rows_per_split,               # 127: split index field
                      (OPTIONS.wgrad_workgroups + tiles - 1) // tiles))


def _wgrad_bf16(dyc: torch.Tensor, xc: torch.Tensor, out_f: int, in_f: int, has_bias: bool):
    """dW [out, in] and db [out] (f32) of a bf16 layer: split-K tiles over the rows, ordered slab sum."""
    from . import lib as L
    so = L.load()
    M = xc.shape[0]
    ksplits = _wgrad_splits(M, ((out_f + 127) // 128) * ((in_f + 127) // 128), 64)
    slab = (out_f * in_f + out_f + 63) // 64 * 64
    arena = torch.empty(ksplits, slab, device=xc.device, dtype=torch.float32)
    db_ptr = arena.data_ptr() + out_f * in_f * 4 if has_bias else None
    L.check(so.mt_wgrad_single_launch(dyc.data_ptr(), xc.data_ptr(), arena.data_ptr(), db_ptr, M, out_f, in_f,
                                      ksplits, slab, L.raw_stream(xc.device)),
            "mt_wgrad_single_launch")
    return _slab_total(so, arena, ksplits, slab, out_f, in_f, has_bias, L.raw_stream(xc.device))


def _wgrad_f32(dyc: torch.Tensor, xc: torch.Tensor, out_f: int, in_f: int, has_bias: bool):
    """dW [out, in] and db [out] of an f32 layer: the contraction over the token rows is split so that the launch has
    about ``OPTIONS.wgrad_workgroups`` workgroups (a slice keeps at least 4 K-steps of 32 rows), slices summed in index
    order."""
    from . import lib as L
    so = L.load()
    M = xc.shape[0]
    ksplits = _wgrad_splits(M, ((out_f + 63) // 64) * ((in_f + 63) // 64), 128)
    slab = (out_f * in_f + out_f + 63) // 64 * 64
    arena = torch.empty(ksplits, slab, device=xc.device, dtype=torch.float32)
    stream = L.raw_stream(xc.device)
    L.check(so.mt_wgrad_f32_single_launch(dyc.data_ptr(), xc.data_ptr(), arena.data_ptr(), M, out_f, in_f, ksplits, slab,
                                          1 if has_bias else 0, stream), "mt_wgrad_f32_single_launch")
    return _slab_total(so, arena, ksplits, slab, out_f, in_f, has_bias, stream)


def _slab_total(so, arena, ksplits, slab, out_f, in_f, has_bias, stream):
    from . import lib as L
    if ksplits > 1:       # slices added in index order by one launch (weights and bias together)
        total = torch.empty(slab, device=arena.device, dtype=torch.float32)
        L.check(so.mt_slab_sum_launch(arena.data_ptr(), total.data_ptr(), out_f * in_f + (out_f if has_bias else 0), ksplits,
                                      slab, stream), "mt_slab_sum_launch")
    else:
        total = arena[0]
    dw = total[:out_f * in_f].view(out_f, in_f)
    db = total[out_f * in_f:out_f * in_f + out_f] if has_bias else None
    return dw, db


class DenseBF16(torch.autograd.Function):
    """A dense layer under bf16 autocast, all three GEMMs on the 128x128 bf16 MFMA tile kernel of ``csrc/gemm.hip``.

    ``x`` arrives in bf16; the weight operand is the bf16 shadow the flat optimiser keeps (``w16``) or a cast of the f32
    master; the bias is added in f32 from the master before the output is rounded to bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16=None):
        from . import lib as L
        so = L.load()
        out_f, in_f = weight.shape
        with torch.autocast("cuda", enabled=False):
            wb = w16 if w16 is not None else weight.detach().to(torch.bfloat16)
            xc = x.reshape(-1, in_f)
            if not xc.is_contiguous():
                xc = xc.contiguous()
            if not wb.is_contiguous():
                wb = wb.contiguous()
            bf = None if bias is None else (bias.detach() if bias.dtype == torch.float32 else bias.detach().float())
            y = torch.empty(xc.shape[0], out_f, device=x.device, dtype=torch.bfloat16)
            L.check(so.mt_gemm_single_launch(0, xc.data_ptr(), wb.data_ptr(), y.data_ptr(),
                                             bf.data_ptr() if bf is not None else None, xc.shape[0], out_f, in_f,
                                             in_f, in_f, out_f, 0, 0, L.raw_stream(x.device)), "mt_gemm_single_launch")
        ctx.save_for_backward(xc, wb)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.view(x.shape[:-1] + (out_f,))

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        so = L.load()
        xc, wb = ctx.saved_tensors
        out_f, in_f = wb.shape
        with torch.autocast("cuda", enabled=False):
            dyc = dy.reshape(-1, out_f)
            if dyc.dtype != torch.bfloat16:
                dyc = dyc.to(torch.bfloat16)
            if not dyc.is_contiguous():
                dyc = dyc.contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty(xc.shape[0], in_f, device=dy.device, dtype=torch.bfloat16)
                L.check(so.mt_gemm_single_launch(1, dyc.data_ptr(), wb.data_ptr(), dx.data_ptr(), None, xc.shape[0], in_f,
                                                 out_f, out_f, in_f, in_f, 0, 0, L.raw_stream(dy.device)),
                        "mt_gemm_single_launch")
                dx = dx.view(ctx.x_shape)
            dw, db = _wgrad_bf16(dyc, xc, out_f, in_f, ctx.has_bias)
        return dx, dw, db, None


def _f32_forward(x, weight, bias):
    """y = x W^T + b on the exact-f32 tile kernel; returns (y, x as [rows, in], W contiguous)."""
    from . import lib as L
    so = L.load()
    out_f, in_f = weight.shape
    xc = x.reshape(-1, in_f)
    if not xc.is_contiguous():
        xc = xc.contiguous()
    wc = weight.detach()
    if not wc.is_contiguous():
        wc = wc.contiguous()
    y = torch.empty(xc.shape[0], out_f, device=x.device, dtype=torch.float32)
    L.check(so.mt_gemm_f32_single_launch(0, xc.data_ptr(), wc.data_ptr(), y.data_ptr(),
                                         bias.data_ptr() if bias is not None else None, None, xc.shape[0], out_f, in_f,
                                         in_f, in_f, out_f, 0, L.raw_stream(x.device)), "mt_gemm_f32_single_launch")
    return y.view(x.shape[:-1] + (out_f,)), xc, wc


class DenseF32(torch.autograd.Function):
    """An f32 dense layer on the exact-f32 MFMA tile kernel of ``csrc/gemm_f32.hip`` (forward, input gradient, and weight
    + bias gradient in one launch each)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y, xc, wc = _f32_forward(x, weight, bias)
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        so = L.load()
        xc, wc = ctx.saved_tensors
        out_f, in_f = wc.shape
        dyc = dy.reshape(-1, out_f)
        if dyc.dtype != torch.float32:
            dyc = dyc.float()
        if not dyc.is_contiguous():
            dyc = dyc.contiguous()
        M, dev = xc.shape[0], dy.device
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, in_f, device=dev, dtype=torch.float32)
            L.check(so.mt_gemm_f32_single_launch(1, dyc.data_ptr(), wc.data_ptr(), dx.data_ptr(), None, None, M, in_f, out_f,
                                                 out_f, in_f, in_f, 0, L.raw_stream(dev)), "mt_gemm_f32_single_launch")
            dx = dx.view(ctx.x_shape)
        dw, db = _wgrad_f32(dyc, xc, out_f, in_f, ctx.has_bias)
        return dx, dw, db


class SmallLinear(torch.autograd.Function):
    """A small f32 dense layer: forward on the exact-f32 64-tile kernel, the whole backward in one launch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y, xc, wc = _f32_forward(x, weight, bias)
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        so = L.load()
        xc, wc = ctx.saved_tensors
        out_f, in_f = wc.shape
        dyc = dy.reshape(-1, out_f)
        if not dyc.is_contiguous():
            dyc = dyc.contiguous()
        R, dev = xc.shape[0], xc.device
        dx = torch.empty(R, in_f, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dw = torch.empty(out_f, in_f, device=dev, dtype=torch.float32)
        db = torch.empty(out_f, device=dev, dtype=torch.float32) if ctx.has_bias else None
        L.check(so.mt_small_linear_bwd_tiled_launch(xc.data_ptr(), wc.data_ptr(), dyc.data_ptr(),
                                                    dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                                    db.data_ptr() if db is not None else None, R, in_f, out_f,
                                                    OPTIONS.small_linear_tile, L.raw_stream(dev)),
                "mt_small_linear_bwd_tiled_launch")
        return (dx.view(ctx.x_shape) if dx is not None else None), dw, db
