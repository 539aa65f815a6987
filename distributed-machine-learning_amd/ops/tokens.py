"""Feature tokenizer and summary-token head of the tabular Transformer.

Tokenizer: ``[B, F]`` numbers -> ``[B, F + 1, d]`` tokens.

``tok[b, t] = x[b, t] * weight[t] + bias[t]`` for the F features and ``tok[b, F] = cls``.  On a GPU one kernel per direction
(``csrc/tokens.hip``) writes the f32 residual stream and, when asked, the bf16 copy the first projection reads; the
backward adds the gradients of both outputs and reduces over the batch in a fixed order.  Elsewhere: plain tensor ops.

Head: ``pred[b] = w . relu(LayerNorm(x[b])) + b0`` on the ``[B, d]`` summary tokens, one kernel per direction (plus the
ordered column sums of the parameter-gradient partials) in place of two dozen small launches.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def feature_tokens_reference(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, cls: torch.Tensor) -> torch.Tensor:
    tok = x.unsqueeze(-1) * weight + bias
    return torch.cat([tok, cls.reshape(1, 1, -1).expand(x.size(0), -1, -1).to(tok.dtype)], dim=1)


class FeatureTokens(torch.autograd.Function):
    BATCH_SLICES = 16        # the batch is reduced in this many slices, added in index order

    @staticmethod
    def forward(ctx, x, weight, bias, cls, want16: bool):
        from . import lib as L
        so = L.load()
        B, F = x.shape
        d = weight.shape[1]
        xc, w, b, c = x.contiguous(), weight.contiguous(), bias.contiguous(), cls.contiguous()
        out = torch.empty(B, F + 1, d, device=x.device, dtype=torch.float32)
        out16 = torch.empty(B, F + 1, d, device=x.device, dtype=torch.bfloat16) if want16 else None
        L.check(so.mt_tokens_fwd_launch(xc.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr(),
                                        out16.data_ptr() if want16 else None, B, F, d, L.raw_stream(x.device)),
                "mt_tokens_fwd_launch")
        ctx.save_for_backward(xc)
        ctx.dims = (B, F, d, tuple(cls.shape))
        if want16:
            return out, out16
        return out

    @staticmethod
    def backward(ctx, g32, g16=None):
        from . import lib as L
        so = L.load()
        (xc,) = ctx.saved_tensors
        B, F, d, cls_shape = ctx.dims
        if g32 is None and g16 is None:
            return None, None, None, None, None
        if g32 is not None:
            g32 = g32.float().contiguous()
        if g16 is not None:
            g16 = g16.to(torch.bfloat16).contiguous()
        cols = (F + 1) * d
        ns = max(1, min(FeatureTokens.BATCH_SLICES, B))
        part = torch.empty(ns, 2, cols, device=xc.device, dtype=torch.float32)
        stream = L.raw_stream(xc.device)
        L.check(so.mt_tokens_bwd_launch(g32.data_ptr() if g32 is not None else None,
                                        g16.data_ptr() if g16 is not None else None, xc.data_ptr(), part.data_ptr(), B, F, d,
                                        ns, stream), "mt_tokens_bwd_launch")
        total = torch.empty(2, cols, device=xc.device, dtype=torch.float32)
        L.check(so.mt_slab_sum_launch(part.data_ptr(), total.data_ptr(), 2 * cols, ns, 2 * cols, stream), "mt_slab_sum_launch")
        dbias = total[0, :F * d].view(F, d)
        dcls = total[0, F * d:].reshape(cls_shape)
        dw = total[1, :F * d].view(F, d)
        return None, dw, dbias, dcls, None


def feature_tokens(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, cls: torch.Tensor,
                   want_bf16: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """``(tokens f32, tokens bf16 or None)``."""
    d = weight.shape[1]
    if (x.is_cuda and x.dim() == 2 and d % 4 == 0 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not x.requires_grad):
        from .layernorm_hip import available
        if available():
            with torch.autocast("cuda", enabled=False):
                out = FeatureTokens.apply(x, weight, bias, cls, bool(want_bf16))
            return out if want_bf16 else (out, None)
    tok = feature_tokens_reference(x.float(), weight, bias, cls)
    return tok, (tok.to(torch.bfloat16) if want_bf16 else None)


def cls_head_reference(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, b0: torch.Tensor,
                       eps: float = 1e-5) -> torch.Tensor:
    h = torch.relu(torch.nn.functional.layer_norm(x, (x.shape[-1],), gamma, beta, eps))
    return (h * w.reshape(-1)).sum(-1) + b0.reshape(())


class ClsHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, w, b0, eps: float):
        from . import lib as L
        so = L.load()
        B, d = x.shape
        xc, g, be, wc, bc = x.contiguous(), gamma.contiguous(), beta.contiguous(), w.contiguous().view(-1), b0.contiguous()
        pred = torch.empty(B, device=x.device, dtype=torch.float32)
        stats = torch.empty(B, 2, device=x.device, dtype=torch.float32)
        L.check(so.mt_cls_head_fwd_launch(xc.data_ptr(), g.data_ptr(), be.data_ptr(), wc.data_ptr(), bc.data_ptr(),
                                          pred.data_ptr(), stats.data_ptr(), B, d, float(eps), L.raw_stream(x.device)),
                "mt_cls_head_fwd_launch")
        ctx.save_for_backward(xc, g, be, wc, stats)
        ctx.shapes = (tuple(w.shape), tuple(b0.shape))
        return pred

    @staticmethod
    def backward(ctx, dpred):
        from . import lib as L
        so = L.load()
        xc, g, be, wc, stats = ctx.saved_tensors
        B, d = xc.shape
        dp = dpred.float().contiguous()
        dx = torch.empty_like(xc)
        nblk = int(so.mt_cls_head_bwd_blocks(B))
        part = torch.empty(nblk, 3 * d + 1, device=xc.device, dtype=torch.float32)
        stream = L.raw_stream(xc.device)
        L.check(so.mt_cls_head_bwd_launch(xc.data_ptr(), g.data_ptr(), be.data_ptr(), wc.data_ptr(), stats.data_ptr(),
                                          dp.data_ptr(), dx.data_ptr(), part.data_ptr(), B, d, stream), "mt_cls_head_bwd_launch")
        tot = torch.empty(3 * d + 1, device=xc.device, dtype=torch.float32)
        L.check(so.mt_colsum_launch(part.data_ptr(), tot.data_ptr(), nblk, 3 * d + 1, stream), "mt_colsum_launch")
        w_shape, b_shape = ctx.shapes
        return dx, tot[:d], tot[d:2 * d], tot[2 * d:3 * d].reshape(w_shape), tot[3 * d:].reshape(b_shape), None


def cls_head(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, b0: torch.Tensor,
             eps: float = 1e-5) -> torch.Tensor:
    """``w . relu(LayerNorm(x)) + b0`` per row of ``x`` [B, d]; ``w`` is the [1, d] weight of the output layer."""
    if x.is_cuda and x.dim() == 2 and x.shape[1] <= 512 and x.dtype == torch.float32 and w.numel() == x.shape[1]:
        from .layernorm_hip import available
        if available():
            with torch.autocast("cuda", enabled=False):
                return ClsHead.apply(x, gamma, beta, w, b0, float(eps))
    return cls_head_reference(x.float(), gamma, beta, w, b0, eps)
