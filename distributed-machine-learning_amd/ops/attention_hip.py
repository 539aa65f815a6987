"""Autograd binding of the fused short-sequence attention kernels (``csrc/attention.hip``)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import lib as L

_KIND = {"softmax": 0, "relu": 1}


def available() -> bool:
    if not torch.cuda.is_available():
        return False
    L.load()          # raises if the library is missing on a GPU machine: no silent fallback
    return True


class FusedShortAttention(torch.autograd.Function):
    """q, k: [B, h, S, dk]; v: [B, h, S, dv]; keep: optional 0/1 mask [B, h, S, S]."""

    @staticmethod
    def forward(ctx, q, k, v, kind: str, keep: Optional[torch.Tensor], dropout_p: float):
        so = L.load()
        B, h, S, dk = q.shape
        dv = v.shape[-1]
        qb, kb, vb = (t.to(torch.bfloat16).contiguous() for t in (q, k, v))
        keep_u8 = keep.to(torch.uint8).contiguous() if keep is not None else None
        out = torch.empty(B, h, S, dv, device=q.device, dtype=torch.bfloat16)
        stat = torch.empty(B * h, S, device=q.device, dtype=torch.float32)
        scale = 1.0 / math.sqrt(dk)
        keep_scale = 1.0 / (1.0 - dropout_p) if keep is not None else 1.0
        L.check(so.mt_attn_fwd_launch(qb.data_ptr(), kb.data_ptr(), vb.data_ptr(),
                                      keep_u8.data_ptr() if keep_u8 is not None else None, out.data_ptr(),
                                      stat.data_ptr(), B * h, S, dk, dv, scale, keep_scale, _KIND[kind],
                                      torch.cuda.current_stream(q.device).cuda_stream), "mt_attn_fwd_launch")
        ctx.save_for_backward(qb, kb, vb, stat, keep_u8 if keep_u8 is not None else torch.empty(0, device=q.device))
        ctx.meta = (kind, scale, keep_scale, keep is not None, q.dtype, k.dtype, v.dtype)
        return out.to(q.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        so = L.load()
        qb, kb, vb, stat, keep_u8 = ctx.saved_tensors
        kind, scale, keep_scale, has_keep, qd, kd, vd = ctx.meta
        B, h, S, dk = qb.shape
        dv = vb.shape[-1]
        do = grad_out.to(torch.bfloat16).contiguous()
        dq = torch.empty(B, h, S, dk, device=qb.device, dtype=torch.float32)
        dk_ = torch.empty_like(dq)
        dv_ = torch.empty(B, h, S, dv, device=qb.device, dtype=torch.float32)
        L.check(so.mt_attn_bwd_launch(qb.data_ptr(), kb.data_ptr(), vb.data_ptr(), do.data_ptr(),
                                      keep_u8.data_ptr() if has_keep else None, stat.data_ptr(), dq.data_ptr(),
                                      dk_.data_ptr(), dv_.data_ptr(), B * h, S, dk, dv, scale, keep_scale,
                                      _KIND[kind], torch.cuda.current_stream(qb.device).cuda_stream),
                "mt_attn_bwd_launch")
        return dq.to(qd), dk_.to(kd), dv_.to(vd), None, None, None
