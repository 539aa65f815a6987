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
                                      L.raw_stream(q.device)), "mt_attn_fwd_launch")
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
                                      _KIND[kind], L.raw_stream(qb.device)),
                "mt_attn_bwd_launch")
        return dq.to(qd), dk_.to(kd), dv_.to(vd), None, None, None


class FusedQKVAttention(torch.autograd.Function):
    """Attention on ONE fused projection output ``qkv`` [B, S, 2*h*dk + h*dv] (all queries, all keys, all values, each
    head-major) -> [B, S, h*dv].  The kernels read the three column blocks in place and the backward writes one
    gradient tensor of the same layout, so the projection is a single GEMM in each direction."""

    @staticmethod
    def forward(ctx, qkv, h: int, dk: int, kind: str, dropout_p: float, seed: Optional[torch.Tensor] = None):
        so = L.load()
        qkv = qkv.contiguous()
        B, S, W = qkv.shape
        dv = (W - 2 * h * dk) // h
        f32 = 1 if qkv.dtype == torch.float32 else 0
        es = qkv.element_size()
        out = torch.empty(B, S, h * dv, device=qkv.device, dtype=qkv.dtype)
        stat = torch.empty(B * h, S, device=qkv.device, dtype=torch.float32)
        scale = 1.0 / math.sqrt(dk)
        drops = seed is not None and dropout_p > 0.0
        keep_scale = 1.0 / (1.0 - dropout_p) if drops else 1.0
        seed = seed if drops else None
        base = qkv.data_ptr()
        L.check(so.mt_attn_fwd_strided_launch(base, base + h * dk * es, base + 2 * h * dk * es, None, out.data_ptr(),
                                              stat.data_ptr(), B * h, S, dk, dv, scale, keep_scale, _KIND[kind], h,
                                              W, W, h * dv, f32, seed.data_ptr() if seed is not None else None,
                                              float(dropout_p), L.raw_stream(qkv.device)),
                "mt_attn_fwd_strided_launch")
        ctx.save_for_backward(qkv, stat)
        ctx.meta = (h, dk, dv, kind, scale, keep_scale, f32)
        ctx.seed, ctx.p = seed, float(dropout_p)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        so = L.load()
        qkv, stat = ctx.saved_tensors
        h, dk, dv, kind, scale, keep_scale, f32 = ctx.meta
        B, S, W = qkv.shape
        es = qkv.element_size()
        do = grad_out.to(qkv.dtype).contiguous()
        # bf16 operands: the kernel rounds the gradient to bf16 itself (mode 2) -- no f32 tensor, no cast pass
        dqkv = torch.empty(B, S, W, device=qkv.device, dtype=qkv.dtype)
        base, g = qkv.data_ptr(), dqkv.data_ptr()
        L.check(so.mt_attn_bwd_strided_launch(base, base + h * dk * es, base + 2 * h * dk * es, do.data_ptr(), None,
                                              stat.data_ptr(), g, g + h * dk * es, g + 2 * h * dk * es, B * h, S, dk, dv,
                                              scale, keep_scale, _KIND[kind], h, W, W, h * dv, 1 if f32 else 2,
                                              ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p,
                                              L.raw_stream(qkv.device)), "mt_attn_bwd_strided_launch")
        return dqkv, None, None, None, None, None


class FusedSelfAttention(torch.autograd.Function):
    """Attention straight on projection outputs: ``qk`` [B, S, 2*h*dk] (queries then keys, head-major), ``v`` [B, S, h*dv]
    -> [B, S, h*dv].  f32 or bf16; no head split / merge copies, no dtype casts: the kernels read the strided operands
    in place and write the gradient of ``qk`` as one tensor."""

    @staticmethod
    def forward(ctx, qk, v, h: int, kind: str, keep: Optional[torch.Tensor], dropout_p: float,
                seed: Optional[torch.Tensor] = None):
        so = L.load()
        if v.dtype != qk.dtype:
            v = v.to(qk.dtype)
        qk, v = qk.contiguous(), v.contiguous()      # a column slice of a fused projection is copied here
        B, S, w2 = qk.shape
        dk, dv = w2 // (2 * h), v.shape[-1] // h
        f32 = 1 if qk.dtype == torch.float32 else 0
        es = qk.element_size()
        keep_u8 = keep.to(torch.uint8).contiguous() if keep is not None else None
        out = torch.empty(B, S, h * dv, device=qk.device, dtype=qk.dtype)
        stat = torch.empty(B * h, S, device=qk.device, dtype=torch.float32)
        scale = 1.0 / math.sqrt(dk)
        drops = keep is not None or (seed is not None and dropout_p > 0.0)
        keep_scale = 1.0 / (1.0 - dropout_p) if drops else 1.0
        seed = seed if (keep is None and drops) else None       # attention dropout drawn inside the kernel
        L.check(so.mt_attn_fwd_strided_launch(qk.data_ptr(), qk.data_ptr() + h * dk * es, v.data_ptr(),
                                              keep_u8.data_ptr() if keep_u8 is not None else None, out.data_ptr(),
                                              stat.data_ptr(), B * h, S, dk, dv, scale, keep_scale, _KIND[kind], h,
                                              2 * h * dk, h * dv, h * dv, f32,
                                              seed.data_ptr() if seed is not None else None, float(dropout_p),
                                              L.raw_stream(qk.device)), "mt_attn_fwd_strided_launch")
        ctx.save_for_backward(qk, v, stat, keep_u8 if keep_u8 is not None else torch.empty(0, device=qk.device))
        ctx.meta = (h, kind, scale, keep_scale, keep is not None, f32)
        ctx.seed, ctx.p = seed, float(dropout_p)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        so = L.load()
        qk, v, stat, keep_u8 = ctx.saved_tensors
        h, kind, scale, keep_scale, has_keep, f32 = ctx.meta
        B, S, w2 = qk.shape
        dk, dv = w2 // (2 * h), v.shape[-1] // h
        es = qk.element_size()
        do = grad_out.to(qk.dtype).contiguous()
        dqk = torch.empty(B, S, w2, device=qk.device, dtype=qk.dtype)
        dv_ = torch.empty(B, S, h * dv, device=qk.device, dtype=qk.dtype)
        L.check(so.mt_attn_bwd_strided_launch(qk.data_ptr(), qk.data_ptr() + h * dk * es, v.data_ptr(), do.data_ptr(),
                                              keep_u8.data_ptr() if has_keep else None, stat.data_ptr(), dqk.data_ptr(),
                                              dqk.data_ptr() + h * dk * es, dv_.data_ptr(), B * h, S, dk, dv, scale,
                                              keep_scale, _KIND[kind], h, 2 * h * dk, h * dv, h * dv, 1 if f32 else 2,
                                              ctx.seed.data_ptr() if ctx.seed is not None else None, ctx.p,
                                              L.raw_stream(qk.device)), "mt_attn_bwd_strided_launch")
        return dqk, dv_, None, None, None, None, None
