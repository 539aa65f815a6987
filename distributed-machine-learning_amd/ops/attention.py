"""Short-sequence attention: one fused HIP kernel per direction on a GPU, plain PyTorch elsewhere.

The reference's sequences are 50 or 96 tokens long
(/root/reference/ray-tune-hpo-regression-sample.py:28, /root/reference/ray-tune-hpo-regression.py:446), so a
whole (batch, head) problem -- Q, K, V and the S x S score tile -- fits in one workgroup's LDS and
registers: no flash-style streaming or online softmax is needed.  Two score functions share the
kernels:

* ``"softmax"``: ``P = softmax(Q K^T / sqrt(dk))``                      (``nn.MultiheadAttention`` semantics)
* ``"relu"``   : ``A = relu(Q K^T / sqrt(dk)); P = A / (rowsum(A) + eps)``  (the reference's softmax-free
  "LinearAttention", :113-117, with a row normalisation so the output scale does not grow with S).

``short_attention`` is differentiable.  On a GPU, shapes inside the kernel envelope (S <= 128, head
dims <= 64 and multiples of 8) run ``ops/csrc/attention.hip`` (bf16 MFMA, f32 softmax and
accumulation); anything else -- and every CPU call -- runs ``attention_reference``.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

RELU_EPS = 1e-6
MAX_SEQ = 128
MAX_HEAD_DIM = 64


def attention_reference(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, kind: str = "softmax",
                        keep_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0) -> torch.Tensor:
    """q, k: [..., S, dk]; v: [..., S, dv]; ``keep_mask`` ([..., S, S], 0/1) applies attention dropout."""
    s = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(q.shape[-1]))
    if kind == "softmax":
        p = torch.softmax(s, dim=-1)
    elif kind == "relu":
        a = torch.relu(s)
        p = a / (a.sum(-1, keepdim=True) + RELU_EPS)
    else:
        raise ValueError(f"kind must be 'softmax' or 'relu', got {kind!r}")
    if keep_mask is not None:
        p = p * keep_mask.to(p.dtype) * (1.0 / (1.0 - dropout_p))
    return p @ v


def _hip_ok(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    if not q.is_cuda or q.dim() != 4:
        return False
    s, dk, dv = q.shape[-2], q.shape[-1], v.shape[-1]
    return (s <= MAX_SEQ and dk <= MAX_HEAD_DIM and dv <= MAX_HEAD_DIM and dk % 8 == 0 and dv % 8 == 0
            and k.shape[-2] == s and v.shape[-2] == s)


def short_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, kind: str = "softmax",
                    dropout_p: float = 0.0) -> torch.Tensor:
    """Self-attention over short sequences, [B, h, S, d] in and out."""
    keep = None
    if dropout_p > 0.0:
        shape = q.shape[:-1] + (k.shape[-2],)
        from .elementwise import keep_mask
        keep = keep_mask(shape, dropout_p, q.device)
    if _hip_ok(q, k, v):
        from .attention_hip import FusedShortAttention, available
        if available():
            return FusedShortAttention.apply(q, k, v, kind, keep, dropout_p)
    return attention_reference(q, k, v, kind, keep, dropout_p)


WIDE_HEAD_DIM = 128      # heads of 65..128 elements fit the kernels' LDS budget for sequences of at most WIDE_MAX_SEQ tokens
WIDE_MAX_SEQ = 96


def _kernel_envelope(t: torch.Tensor, S: int, dk: int, dv: int) -> bool:
    """Strided (projection-output) operands: any head width from 1 to 64 -- widths that are not multiples of 8 are staged
    element by element -- and up to 128 for short sequences."""
    if not (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and S <= MAX_SEQ and dk > 0 and dv > 0):
        return False
    widest = max(dk, dv)
    return widest <= MAX_HEAD_DIM or (widest <= WIDE_HEAD_DIM and S <= WIDE_MAX_SEQ)


def self_attention_qkv(qkv: torch.Tensor, num_heads: int, dk: int, kind: str = "softmax",
                       dropout_p: float = 0.0) -> torch.Tensor:
    """Multi-head self-attention on ONE fused projection output ``qkv`` [B, S, 2*h*dk + h*dv]: all queries, all keys,
    all values, each block head-major.  On a GPU the kernels read the three blocks in place and return one gradient
    tensor for the projection; elsewhere the blocks are split and ``self_attention`` runs."""
    B, S, W = qkv.shape
    h = num_heads
    dv = (W - 2 * h * dk) // h
    if _kernel_envelope(qkv, S, dk, dv):
        from .attention_hip import FusedQKVAttention, available
        if available():
            seed = None
            if dropout_p > 0.0:
                from .rng import next_seed
                seed = next_seed(qkv.device)
            return FusedQKVAttention.apply(qkv, h, dk, kind, dropout_p, seed)
    qk, v = qkv.split([2 * h * dk, h * dv], dim=-1)
    return self_attention(qk, v, h, kind, dropout_p)


def self_attention(qk: torch.Tensor, v: torch.Tensor, num_heads: int, kind: str = "softmax",
                   dropout_p: float = 0.0) -> torch.Tensor:
    """Multi-head self-attention on projection outputs: ``qk`` [B, S, 2*h*dk] (all queries, then all keys, head-major),
    ``v`` [B, S, h*dv]; returns [B, S, h*dv].  On a GPU the fused kernels read these tensors in place (no head
    split/merge copies, no casts); elsewhere the heads are split and ``attention_reference`` runs."""
    B, S, w2 = qk.shape
    h = num_heads
    dk, dv = w2 // (2 * h), v.shape[-1] // h
    if _kernel_envelope(qk, S, dk, dv):
        from .attention_hip import FusedSelfAttention, available
        if available():
            seed = None
            if dropout_p > 0.0:           # no [B, h, S, S] mask tensor: the kernels hash (seed, element index)
                from .rng import next_seed
                seed = next_seed(qk.device)
            return FusedSelfAttention.apply(qk, v, h, kind, None, dropout_p, seed)
    keep = None
    if dropout_p > 0.0:
        from .elementwise import keep_mask
        keep = keep_mask((B, h, S, S), dropout_p, qk.device)
    q, k = qk.chunk(2, dim=-1)
    split = lambda t, d: t.reshape(B, S, h, d).transpose(1, 2)          # noqa: E731
    o = attention_reference(split(q, dk), split(k, dk), split(v, dv), kind, keep, dropout_p)
    return o.transpose(1, 2).reshape(B, S, h * dv)
