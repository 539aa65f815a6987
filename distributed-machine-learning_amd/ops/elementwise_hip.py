"""Autograd bindings of ``csrc/posenc.hip`` and ``csrc/conv1d.hip``."""
from __future__ import annotations

import torch

from . import lib as L


def _stream(t: torch.Tensor) -> int:
    return L.raw_stream(t.device)


class FusedPosEncDropout(torch.autograd.Function):
    """``dropout(x + pe[:S])`` with the keep mask recomputed from a device-resident seed in backward."""

    @staticmethod
    def forward(ctx, x, pe, seed, p: float):
        so = L.load()
        B, S, d = x.shape
        if pe.shape[0] < S or pe.shape[1] != d:
            raise ValueError(f"positional table {tuple(pe.shape)} does not cover a [{S}, {d}] sequence")
        xc, pec = x.contiguous(), pe.contiguous().float()
        y = torch.empty_like(xc)
        L.check(so.mt_posenc_fwd_launch(xc.data_ptr(), pec.data_ptr(), y.data_ptr(), B * S, S, d,
                                        seed.data_ptr() if seed is not None else None, float(p), _stream(x)),
                "mt_posenc_fwd_launch")
        ctx.seed, ctx.p = seed, float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        so = L.load()
        dyc = dy.contiguous()
        B, S, d = dyc.shape
        dx = torch.empty_like(dyc)
        L.check(so.mt_posenc_bwd_launch(dyc.data_ptr(), dx.data_ptr(), B * S, d,
                                        ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p, _stream(dy)),
                "mt_posenc_bwd_launch")
        return dx, None, None, None


class FusedDepthwiseConv1d(torch.autograd.Function):
    """Depthwise conv over time on ``[B, S, d]`` without transposes; weight is ``[d, 1, k]`` or ``[d, k]``."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        so = L.load()
        B, S, d = x.shape
        k = weight.shape[-1]
        xc = x.contiguous()
        w = weight.reshape(d, k).contiguous()
        y = torch.empty_like(xc)
        L.check(so.mt_dwconv_launch(xc.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    y.data_ptr(), B * S, S, d, k, 0, _stream(x)), "mt_dwconv_launch")
        ctx.save_for_backward(xc, w)
        ctx.meta = (weight.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        so = L.load()
        xc, w = ctx.saved_tensors
        wshape, has_bias = ctx.meta
        B, S, d = xc.shape
        k = w.shape[1]
        dyc = dy.contiguous()
        dx = torch.empty_like(xc)
        L.check(so.mt_dwconv_launch(dyc.data_ptr(), w.data_ptr(), None, dx.data_ptr(), B * S, S, d, k, 1, _stream(dy)),
                "mt_dwconv_launch(dx)")
        chunks = int(so.mt_dwconv_chunks(B * S))
        part = torch.empty(chunks, k + 1, d, device=xc.device, dtype=torch.float32)
        L.check(so.mt_dwconv_wgrad_launch(xc.data_ptr(), dyc.data_ptr(), part.data_ptr(), B * S, S, d, k, _stream(dy)),
                "mt_dwconv_wgrad_launch")
        g = torch.empty(k + 1, d, device=xc.device, dtype=torch.float32)     # fixed summation order
        L.check(so.mt_colsum_launch(part.data_ptr(), g.data_ptr(), chunks, (k + 1) * d, _stream(dy)), "mt_colsum_launch")
        dw = g[:k].t().reshape(wshape)
        return dx, dw, (g[k] if has_bias else None)


def hip_keep_mask(shape, p: float, device) -> torch.Tensor:
    from .rng import next_seed
    so = L.load()
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    seed = next_seed(device)
    L.check(so.mt_keep_mask_launch(out.data_ptr(), out.numel(), seed.data_ptr(), p,
                                   L.raw_stream(device)), "mt_keep_mask_launch")
    return out


class FusedDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seed, p: float):
        so = L.load()
        xc = x.contiguous()
        y = torch.empty_like(xc)
        L.check(so.mt_dropout_launch(xc.data_ptr(), y.data_ptr(), xc.numel(), seed.data_ptr(), p,
                                     0 if x.dtype == torch.float32 else 1, _stream(x)), "mt_dropout_launch")
        ctx.seed, ctx.p = seed, p
        return y

    @staticmethod
    def backward(ctx, dy):
        so = L.load()
        dyc = dy.contiguous()
        dx = torch.empty_like(dyc)
        L.check(so.mt_dropout_launch(dyc.data_ptr(), dx.data_ptr(), dyc.numel(), ctx.seed.data_ptr(), ctx.p,
                                     0 if dy.dtype == torch.float32 else 1, _stream(dy)), "mt_dropout_launch(bwd)")
        return dx, None, None


class FusedReluDropout(torch.autograd.Function):
    """``dropout(relu(x))`` in one pass; the output is the only tensor kept for backward (non-zero = positive and kept)."""

    @staticmethod
    def forward(ctx, x, seed, p: float):
        so = L.load()
        xc = x.contiguous()
        y = torch.empty_like(xc)
        L.check(so.mt_relu_dropout_fwd_launch(xc.data_ptr(), y.data_ptr(), xc.numel(), seed.data_ptr(), p,
                                              0 if x.dtype == torch.float32 else 1, _stream(x)), "mt_relu_dropout_fwd_launch")
        ctx.save_for_backward(y)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        so = L.load()
        (y,) = ctx.saved_tensors
        dyc = dy.to(y.dtype).contiguous()
        dx = torch.empty_like(dyc)
        L.check(so.mt_relu_dropout_bwd_launch(dyc.data_ptr(), y.data_ptr(), dx.data_ptr(), dyc.numel(), ctx.p,
                                              0 if y.dtype == torch.float32 else 1, _stream(dy)), "mt_relu_dropout_bwd_launch")
        return dx, None, None


class FusedReGLUDropout(torch.autograd.Function):
    """``dropout(a * relu(g))`` for ``u = [a | g]`` in bf16: one launch per direction (``csrc/dropout.hip``); the backward
    writes both halves of ``du`` in place of five element-wise launches and a concatenation."""

    @staticmethod
    def forward(ctx, u, seed, p: float):
        so = L.load()
        h = u.shape[-1] // 2
        uc = u.contiguous()
        rows = uc.numel() // (2 * h)
        out = torch.empty(u.shape[:-1] + (h,), device=u.device, dtype=torch.bfloat16)
        L.check(so.mt_reglu_fwd_launch(uc.data_ptr(), out.data_ptr(), rows, h, seed.data_ptr() if seed is not None else None,
                                       float(p), _stream(u)), "mt_reglu_fwd_launch")
        ctx.save_for_backward(uc)
        ctx.seed, ctx.p, ctx.h, ctx.rows = seed, float(p), h, rows
        return out

    @staticmethod
    def backward(ctx, dout):
        so = L.load()
        (uc,) = ctx.saved_tensors
        dc = dout.to(torch.bfloat16).contiguous()
        du = torch.empty_like(uc)
        L.check(so.mt_reglu_bwd_launch(uc.data_ptr(), dc.data_ptr(), du.data_ptr(), ctx.rows, ctx.h,
                                       ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p, _stream(dout)),
                "mt_reglu_bwd_launch")
        return du, None, None
