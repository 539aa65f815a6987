"""Feature tokenizer of the tabular Transformer: ``[B, F]`` numbers -> ``[B, F + 1, d]`` tokens.

``tok[b, t] = x[b, t] * weight[t] + bias[t]`` for the F features and ``tok[b, F] = cls``.  On a GPU one kernel per direction
(``csrc/tokens.hip``) writes the f32 residual stream and, when asked, the bf16 copy the first projection reads; the
backward adds the gradients of both outputs and reduces over the batch in a fixed order.  Elsewhere: plain tensor ops.
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
