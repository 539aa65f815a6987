"""Dense layers whose backward runs on hand-written MFMA kernels.

``linear(x, weight, bias)`` is ``F.linear`` everywhere except in two places.

Small f32 layers (no autocast; at most ``SMALL_MAX_ROWS`` rows and ``in x out <= SMALL_MAX_WEIGHTS`` -- every
projection of the sequence-regression Transformers at batch 32 x 50 tokens): the backward's three library launches
(input gradient, weight gradient, bias column sum) become ONE launch of ``mt_small_linear_bwd_tiled_launch``
(``csrc/small_linear.hip``, exact-f32 matrix instructions, sixteen waves per 32x32 tile adding their accumulators
through LDS in a fixed order; no workspace and no traffic between workgroups).  Measured per backward inside a replayed graph (library / fused, us): 800x64x256 61.5 / 6.4, 1600x10x32 20.5 / 6.6,
1600x64x256 29.9 / 6.9, 1600x128x384 27.5 / 13.0, 1600x128x512 28.3 / 15.3, 3200x64x256 43.7 / 18.3; past the bound the
library's larger LDS-staged tiles catch up (4800x256x768 77.1 / 74.4, 1600x256x1024 37.4 / 40.5) and keep the layer.

Everything else on a GPU runs on the tile kernels of ``csrc/gemm.hip`` (bf16 autocast: ``DenseBF16``) and
``csrc/gemm_f32.hip`` (f32: ``DenseF32``): forward with the bias in the epilogue, input gradient, and the weight + bias
gradient ``dW = dY^T X`` -- a long skinny contraction over the token rows (M = batch x tokens ~ 10^4..10^5 against a few
hundred features) -- split over enough workgroups to fill the chip, the f32 partial slabs added in a fixed order by one
small launch.  No BLAS-library GEMM is left in a training step (``MITUNE_OWN_GEMM=0`` restores the library for
comparison).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

SPLIT_TARGET = int(os.environ.get("MITUNE_WGRAD_WORKGROUPS", "512"))      # workgroups a weight-gradient launch aims for
MIN_ROWS = 4096        # below this the library GEMM is fine and the slab reduction is pure overhead
SMALL_MAX_ROWS = 32768
SMALL_MAX_WEIGHTS = 65536     # in x out, e.g. 128 x 512
SMALL_MAX_MACS = 1 << 29      # rows x in x out; measured wins up to here: 6144x128x512 66.5 / 51.7 us, 12288x64x192
                              # 109.4 / 58.3, 24576x64x128 135.8 / 110.4 (library / fused)


def _eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (x.is_cuda and torch.is_autocast_enabled() and torch.is_grad_enabled() and weight.requires_grad):
        return False
    if torch.get_autocast_dtype("cuda") != torch.bfloat16:
        return False
    out_f, in_f = weight.shape
    rows = x.numel() // max(in_f, 1)
    return rows >= MIN_ROWS and out_f % 8 == 0 and in_f % 8 == 0 and out_f <= 128 * 255


def _small_eligible(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    if not (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and not torch.is_autocast_enabled()):
        return False
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        return False
    out_f, in_f = weight.shape
    rows = x.numel() // max(in_f, 1)
    return 0 < rows <= SMALL_MAX_ROWS and in_f * out_f <= SMALL_MAX_WEIGHTS and rows * in_f * out_f <= SMALL_MAX_MACS


def _shadow(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bf16 copy of a parameter kept current by the flat optimiser (``FlatParams(bf16_shadow=True)``), if there is one."""
    return getattr(t, "_mt_bf16", None) if t is not None else None


OWN_GEMM = os.environ.get("MITUNE_OWN_GEMM", "1") != "0"      # 0: dense forward / input gradient back on the BLAS library


def _bf16_autocast() -> bool:
    return torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``x @ weight.T + bias`` on this package's own MFMA kernels whenever the operands allow 16-byte rows, so that a
    training step launches no BLAS-library GEMM at all: forward, input gradient and weight gradient of bf16-autocast
    layers on the bf16 tile kernel (``DenseBF16``), of f32 layers on the exact-f32 tile kernel (``DenseF32``) or, for
    small f32 layers, on the fused one-launch backward (``SmallLinear``)."""
    out_f, in_f = weight.shape
    if x.is_cuda and OWN_GEMM and x.numel() > 0:
        if out_f == 1 and torch.is_autocast_enabled():
            # a scalar head under autocast: a row-wise dot product in f32 (one multiply + one row reduction), not a GEMV
            with torch.autocast("cuda", enabled=False):
                y = (x.float() * weight.float().reshape(-1)).sum(-1, keepdim=True)
                return y if bias is None else y + bias.float()
        if _bf16_autocast() and out_f % 8 == 0 and in_f % 8 == 0 and weight.dtype == torch.float32:
            xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
            return DenseBF16.apply(xb, weight, bias, _shadow(weight))
        if (not torch.is_autocast_enabled() and x.dtype == torch.float32 and weight.dtype == torch.float32
                and (bias is None or bias.dtype == torch.float32)):
            if _small_eligible(x, weight, bias):
                return SmallLinear.apply(x, weight, bias)
            if out_f % 4 == 0 and in_f % 4 == 0:
                return DenseF32.apply(x, weight, bias)
            if out_f % 4 == 0 and torch.is_grad_enabled() and x.numel() // in_f >= MIN_ROWS:
                # a big layer whose rows are not 16-byte aligned (81 input features): zero columns up to the next
                # multiple of 4 on both operands keep it on the tile kernels; autograd slices the weight gradient back
                pad = (-in_f) % 4
                return DenseF32.apply(F.pad(x, (0, pad)), F.pad(weight, (0, pad)), bias)
            if _small_inference(x, weight, bias):
                return SmallLinear.forward(_NoCtx(), x, weight, bias)
    if _eligible(x, weight):
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        return SplitKLinear.apply(xb, weight, bias, _shadow(weight), _shadow(bias))
    w16 = _shadow(weight)
    if (w16 is not None and x.is_cuda and not torch.is_grad_enabled() and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and (bias is None or _shadow(bias) is not None)):
        with torch.autocast("cuda", enabled=False):       # validation pass: no per-call casts of the weights either
            return F.linear(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), w16, _shadow(bias))
    if _small_eligible(x, weight, bias):
        return SmallLinear.apply(x, weight, bias)
    if _small_inference(x, weight, bias):
        return SmallLinear.forward(_NoCtx(), x, weight, bias)
    return F.linear(x, weight, bias)


def _wgrad_bf16(dyc: torch.Tensor, xc: torch.Tensor, out_f: int, in_f: int, has_bias: bool):
    """dW [out, in] and db [out] (f32) of a bf16 layer: split-K tiles over the rows, ordered slab sum."""
    from . import lib as L
    so = L.load()
    M = xc.shape[0]
    tiles = ((out_f + 127) // 128) * ((in_f + 127) // 128)
    ksplits = max(1, min(127, (M + 63) // 64, (SPLIT_TARGET + tiles - 1) // tiles))     # 127: split index field
    slab = (out_f * in_f + out_f + 63) // 64 * 64
    arena = torch.empty(ksplits, slab, device=xc.device, dtype=torch.float32)
    db_ptr = arena.data_ptr() + out_f * in_f * 4 if has_bias else None
    L.check(so.mt_wgrad_single_launch(dyc.data_ptr(), xc.data_ptr(), arena.data_ptr(), db_ptr, M, out_f, in_f,
                                      ksplits, slab, L.raw_stream(xc.device)),
            "mt_wgrad_single_launch")
    if ksplits > 1:       # slices added in index order by one launch (weights and bias together)
        total = torch.empty(slab, device=xc.device, dtype=torch.float32)
        L.check(so.mt_slab_sum_launch(arena.data_ptr(), total.data_ptr(), out_f * in_f + (out_f if has_bias else 0), ksplits,
                                      slab, L.raw_stream(xc.device)), "mt_slab_sum_launch")
    else:
        total = arena[0]
    dw = total[:out_f * in_f].view(out_f, in_f)
    db = total[out_f * in_f:out_f * in_f + out_f] if has_bias else None
    return dw, db


def _wgrad_f32(dyc: torch.Tensor, xc: torch.Tensor, out_f: int, in_f: int, has_bias: bool):
    """dW [out, in] and db [out] of an f32 layer: the contraction over the token rows is split so that the launch has
    about ``SPLIT_TARGET`` workgroups (a slice keeps at least 4 K-steps of 32 rows), slices summed in index order."""
    from . import lib as L
    so = L.load()
    M = xc.shape[0]
    tiles = ((out_f + 63) // 64) * ((in_f + 63) // 64)
    ksplits = max(1, min(127, (M + 127) // 128, (SPLIT_TARGET + tiles - 1) // tiles))
    slab = (out_f * in_f + out_f + 63) // 64 * 64
    arena = torch.empty(ksplits, slab, device=xc.device, dtype=torch.float32)
    stream = L.raw_stream(xc.device)
    L.check(so.mt_wgrad_f32_single_launch(dyc.data_ptr(), xc.data_ptr(), arena.data_ptr(), M, out_f, in_f, ksplits, slab,
                                          1 if has_bias else 0, stream), "mt_wgrad_f32_single_launch")
    if ksplits > 1:
        total = torch.empty(slab, device=xc.device, dtype=torch.float32)
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


class DenseF32(torch.autograd.Function):
    """An f32 dense layer on the exact-f32 MFMA tile kernel of ``csrc/gemm_f32.hip`` (forward, input gradient, and weight
    + bias gradient in one launch each)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
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
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.view(x.shape[:-1] + (out_f,))

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


class _NoCtx:
    def save_for_backward(self, *a):
        pass


def _small_inference(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """The same small f32 layers when no gradient is recorded (validation passes): the forward kernel alone."""
    if torch.is_grad_enabled() or not x.is_cuda or torch.is_autocast_enabled() or not (SmallLinear.OWN_FORWARD or OWN_GEMM):
        return False
    if x.dtype != torch.float32 or weight.dtype != torch.float32 or (bias is not None and bias.dtype != torch.float32):
        return False
    out_f, in_f = weight.shape
    rows = x.numel() // max(in_f, 1)
    return 0 < rows <= 4 * SMALL_MAX_ROWS and in_f * out_f <= SMALL_MAX_WEIGHTS


class SmallLinear(torch.autograd.Function):
    TILE = int(os.environ.get("MITUNE_SMALL_LINEAR_TILE", "0"))   # 0: the launcher chooses 16x16 or 32x32 output tiles by layer size; 16 / 32 force one (tests, ablations)

    # The forward has a kernel of the same family too (mt_small_linear_fwd_launch), OFF by default: measured against the
    # library GEMM it is slower alone (1600x64x256: 7.5 vs 4.2 us, 1600x128x512: 11.0 vs 5.8 us) and, with its
    # 1024-thread workgroups, it nearly halves what four concurrently replayed trials gain (step 110 -> 196 us).
    OWN_FORWARD = bool(int(os.environ.get("MITUNE_SMALL_LINEAR_FWD", "0")))

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        from . import lib as L
        out_f, in_f = weight.shape
        if OWN_GEMM and out_f % 4 == 0 and in_f % 4 == 0:       # 16-byte rows: the 64-tile exact-f32 MFMA kernel
            so = L.load()
            xc = x.reshape(-1, in_f).contiguous()
            wc = weight.detach().contiguous()
            y = torch.empty(xc.shape[0], out_f, device=x.device, dtype=torch.float32)
            L.check(so.mt_gemm_f32_single_launch(0, xc.data_ptr(), wc.data_ptr(), y.data_ptr(),
                                                 bias.data_ptr() if bias is not None else None, None, xc.shape[0], out_f,
                                                 in_f, in_f, in_f, out_f, 0, L.raw_stream(x.device)),
                    "mt_gemm_f32_single_launch")
            return y.view(x.shape[:-1] + (out_f,))
        if not (SmallLinear.OWN_FORWARD or OWN_GEMM):
            return F.linear(x, weight, bias)
        so = L.load()
        xc = x.reshape(-1, in_f).contiguous()
        wc = weight.contiguous()
        y = torch.empty(xc.shape[0], out_f, device=x.device, dtype=torch.float32)
        L.check(so.mt_small_linear_fwd_launch(xc.data_ptr(), wc.data_ptr(), bias.data_ptr() if bias is not None else None,
                                              y.data_ptr(), xc.shape[0], in_f, out_f, 0,
                                              L.raw_stream(x.device)),
                "mt_small_linear_fwd_launch")
        return y.view(x.shape[:-1] + (out_f,))

    @staticmethod
    def backward(ctx, dy):
        from . import lib as L
        so = L.load()
        x, weight = ctx.saved_tensors
        out_f, in_f = weight.shape
        dyc = dy.reshape(-1, out_f).contiguous()
        xc = x.reshape(-1, in_f).contiguous()
        R = xc.shape[0]
        dev = x.device
        dx = torch.empty(R, in_f, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dw = torch.empty(out_f, in_f, device=dev, dtype=torch.float32)
        db = torch.empty(out_f, device=dev, dtype=torch.float32) if ctx.has_bias else None
        L.check(so.mt_small_linear_bwd_tiled_launch(xc.data_ptr(), weight.data_ptr(), dyc.data_ptr(),
                                                    dx.data_ptr() if dx is not None else None, dw.data_ptr(),
                                                    db.data_ptr() if db is not None else None, R, in_f, out_f,
                                                    SmallLinear.TILE, L.raw_stream(dev)),
                "mt_small_linear_bwd_tiled_launch")
        return (dx.reshape(x.shape) if dx is not None else None), dw, db


class SplitKLinear(torch.autograd.Function):
    """``x`` arrives in bf16; ``w16`` / ``b16`` are bf16 shadows of the f32 master parameters when the optimiser keeps
    them (one cast of the whole parameter slab per step instead of two cast launches per layer call), else None."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16=None, b16=None):
        with torch.autocast("cuda", enabled=False):
            wb = w16 if w16 is not None else weight.to(torch.bfloat16)
            bb = None if bias is None else (b16 if b16 is not None else bias.to(torch.bfloat16))
            ctx.save_for_backward(x, wb)
            ctx.has_bias = bias is not None
            return F.linear(x, wb, bb)

    @staticmethod
    def backward(ctx, dy):
        with torch.autocast("cuda", enabled=False):
            return SplitKLinear._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        from . import lib as L
        so = L.load()
        x, weight = ctx.saved_tensors
        out_f, in_f = weight.shape
        dyc = dy.reshape(-1, out_f).to(torch.bfloat16).contiguous()
        xc = x.reshape(-1, in_f).contiguous()
        M = xc.shape[0]
        dx = (dyc @ weight).reshape(x.shape) if ctx.needs_input_grad[0] else None
        tiles = ((out_f + 127) // 128) * ((in_f + 127) // 128)
        ksplits = max(1, min(127, (M + 63) // 64, (SPLIT_TARGET + tiles - 1) // tiles))     # 127: split index field
        slab = (out_f * in_f + out_f + 63) // 64 * 64
        arena = torch.empty(ksplits, slab, device=x.device, dtype=torch.float32)
        db_ptr = arena.data_ptr() + out_f * in_f * 4 if ctx.has_bias else None
        L.check(so.mt_wgrad_single_launch(dyc.data_ptr(), xc.data_ptr(), arena.data_ptr(), db_ptr, M, out_f, in_f,
                                          ksplits, slab, L.raw_stream(x.device)),
                "mt_wgrad_single_launch")
        total = arena.sum(0) if ksplits > 1 else arena[0]
        dw = total[:out_f * in_f].view(out_f, in_f)
        db = total[out_f * in_f:out_f * in_f + out_f] if ctx.has_bias else None
        return dx, dw, db, None, None
