"""Autograd binding of the fused dropout + residual + LayerNorm kernels (``csrc/layernorm.hip``)."""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as L


def available() -> bool:
    if not torch.cuda.is_available():
        return False
    L.load()
    return True


class FusedResidualLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps: float, keep: Optional[torch.Tensor], dropout_p: float,
                seed: Optional[torch.Tensor] = None):
        """Dropout on ``x``: a stored 0/1 ``keep`` mask, or (``seed`` [1] int64 on the device) a mask drawn inside the
        kernel and re-drawn from the same seed in backward; neither: no dropout."""
        so = L.load()
        d = x.shape[-1]
        xf, rf = x.contiguous().view(-1, d), residual.contiguous().view(-1, d)
        M = xf.shape[0]
        keep_u8 = keep.to(torch.uint8).contiguous() if keep is not None else None
        y, z = torch.empty_like(xf), torch.empty_like(xf)
        stats = torch.empty(M, 2, device=x.device, dtype=torch.float32)
        drops = keep is not None or (seed is not None and dropout_p > 0.0)
        keep_scale = 1.0 / (1.0 - dropout_p) if drops else 1.0
        seed = seed if (keep is None and drops) else None
        w, b = weight.contiguous(), bias.contiguous()
        L.check(so.mt_ln_fwd_launch(xf.data_ptr(), rf.data_ptr(), keep_u8.data_ptr() if keep_u8 is not None else None,
                                    w.data_ptr(), b.data_ptr(), y.data_ptr(), z.data_ptr(), stats.data_ptr(), M, d,
                                    float(eps), keep_scale, seed.data_ptr() if seed is not None else None,
                                    float(dropout_p), L.raw_stream(x.device)),
                "mt_ln_fwd_launch")
        ctx.save_for_backward(z, stats, w, keep_u8 if keep_u8 is not None else torch.empty(0, device=x.device))
        ctx.meta = (keep is not None, keep_scale, x.shape)
        ctx.seed, ctx.p = seed, float(dropout_p)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, grad_y):
        so = L.load()
        z, stats, w, keep_u8 = ctx.saved_tensors
        has_keep, keep_scale, shape = ctx.meta
        M, d = z.shape
        dy = grad_y.contiguous().view(M, d)
        dx, dres = torch.empty_like(z), torch.empty_like(z)
        nblk = int(so.mt_ln_bwd_blocks(M))
        part = torch.empty(nblk, 2, d, device=z.device, dtype=torch.float32)
        L.check(so.mt_ln_bwd_launch(dy.data_ptr(), z.data_ptr(), stats.data_ptr(),
                                    keep_u8.data_ptr() if has_keep else None, w.data_ptr(), dx.data_ptr(),
                                    dres.data_ptr(), part.data_ptr(), M, d, keep_scale,
                                    ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p,
                                    L.raw_stream(z.device)), "mt_ln_bwd_launch")
        dwb = torch.empty(2, d, device=z.device, dtype=torch.float32)
        L.check(so.mt_colsum_launch(part.data_ptr(), dwb.data_ptr(), nblk, 2 * d, L.raw_stream(z.device)), "mt_colsum_launch")
        return dx.view(shape), dres.view(shape), dwb[0], dwb[1], None, None, None, None


class MixedResidualLayerNorm(torch.autograd.Function):
    """``LayerNorm(residual + dropout(x))`` for bf16-autocast models (rows of at most 256 columns): ``x`` in bf16 or f32,
    ``residual`` f32; returns the f32 output and, when asked, a bf16 copy of it for the next GEMM -- no cast kernel on
    either side.  Backward adds the gradients of the two outputs inside the kernel and returns the gradient of ``x`` in
    ``x``'s precision."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps: float, dropout_p: float, seed: Optional[torch.Tensor], want16: bool):
        so = L.load()
        d = x.shape[-1]
        xf, rf = x.contiguous().view(-1, d), residual.contiguous().view(-1, d)
        M = xf.shape[0]
        xb = xf.dtype == torch.bfloat16
        y, z = torch.empty_like(rf), torch.empty_like(rf)
        y16 = torch.empty(M, d, device=x.device, dtype=torch.bfloat16) if want16 else None
        stats = torch.empty(M, 2, device=x.device, dtype=torch.float32)
        drops = seed is not None and dropout_p > 0.0
        keep_scale = 1.0 / (1.0 - dropout_p) if drops else 1.0
        seed = seed if drops else None
        w, b = weight.contiguous(), bias.contiguous()
        L.check(so.mt_ln_fwd_mixed_launch(xf.data_ptr(), int(xb), rf.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                          y16.data_ptr() if y16 is not None else None, z.data_ptr(), stats.data_ptr(), M, d,
                                          float(eps), keep_scale, seed.data_ptr() if seed is not None else None,
                                          float(dropout_p), L.raw_stream(x.device)), "mt_ln_fwd_mixed_launch")
        ctx.save_for_backward(z, stats, w)
        ctx.meta = (keep_scale, x.shape, xb, want16)
        ctx.seed, ctx.p = seed, float(dropout_p)
        if want16:
            return y.view(x.shape), y16.view(x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, grad_y, grad_y16=None):
        so = L.load()
        z, stats, w = ctx.saved_tensors
        keep_scale, shape, xb, _ = ctx.meta
        M, d = z.shape
        dy = None if grad_y is None else grad_y.contiguous().view(M, d)
        if dy is not None and dy.dtype != torch.float32:
            dy = dy.float()
        dy16 = None if grad_y16 is None else grad_y16.contiguous().view(M, d)
        if dy16 is not None and dy16.dtype != torch.bfloat16:
            dy16 = dy16.to(torch.bfloat16)
        if dy is None and dy16 is None:
            return (None,) * 8
        dx = torch.empty(M, d, device=z.device, dtype=torch.bfloat16 if xb else torch.float32)
        dres = torch.empty_like(z)
        nblk = int(so.mt_ln_bwd_blocks(M))
        part = torch.empty(nblk, 2, d, device=z.device, dtype=torch.float32)
        L.check(so.mt_ln_bwd_mixed_launch(dy.data_ptr() if dy is not None else None,
                                          dy16.data_ptr() if dy16 is not None else None, z.data_ptr(), stats.data_ptr(),
                                          w.data_ptr(), dx.data_ptr(), int(xb), dres.data_ptr(), part.data_ptr(), M, d,
                                          keep_scale, ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p,
                                          L.raw_stream(z.device)), "mt_ln_bwd_mixed_launch")
        dwb = torch.empty(2, d, device=z.device, dtype=torch.float32)
        L.check(so.mt_colsum_launch(part.data_ptr(), dwb.data_ptr(), nblk, 2 * d, L.raw_stream(z.device)), "mt_colsum_launch")
        return dx.view(shape), dres.view(shape), dwb[0], dwb[1], None, None, None, None
