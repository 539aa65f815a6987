"""Small fused ops of the Transformer path: positional-encoding add + dropout, channel-last depthwise conv1d.

``pe_add_dropout``  <-> /root/reference/ray-tune-hpo-regression.py:53-54 (``x + pe[:S]`` then dropout)
``depthwise_conv1d`` <-> /root/reference/ray-tune-hpo-regression.py:68,79-82 (``Conv1d(d, d, k, groups=d)``
between two transposes)

On a GPU both run as hand-written kernels (``csrc/posenc.hip``, ``csrc/conv1d.hip``) that work on the
``[B, S, d]`` layout the encoder keeps, so no transpose and no mask tensor is ever materialised; elsewhere
the plain PyTorch forms below are used (they are also the numerics references of the kernel tests).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def pe_add_dropout_reference(x: torch.Tensor, pe: torch.Tensor, keep: Optional[torch.Tensor] = None, p: float = 0.0):
    y = x + pe[: x.shape[1]].to(x.dtype)
    if keep is not None:
        y = y * keep.to(y.dtype) / (1.0 - p)
    return y


def pe_add_dropout(x: torch.Tensor, pe: torch.Tensor, p: float = 0.0, training: bool = False) -> torch.Tensor:
    p = float(p) if training else 0.0
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0:
        from .elementwise_hip import FusedPosEncDropout
        from .rng import next_seed
        return FusedPosEncDropout.apply(x, pe, next_seed(x.device) if p > 0.0 else None, p)
    return F.dropout(x + pe[: x.shape[1]].to(x.dtype), p, training=p > 0.0)


def depthwise_conv1d_reference(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x [B, S, d]; weight [d, 1, k] (the ``nn.Conv1d(groups=d)`` parameter) or [d, k]; returns [B, S, d]."""
    w = weight.reshape(weight.shape[0], 1, -1)
    return F.conv1d(x.transpose(1, 2), w, bias, padding=w.shape[-1] // 2, groups=w.shape[0]).transpose(1, 2)


def depthwise_conv1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    k = weight.shape[-1]
    if x.is_cuda and x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and k % 2 == 1 and k <= 7:
        from .elementwise_hip import FusedDepthwiseConv1d
        return FusedDepthwiseConv1d.apply(x, weight, bias)
    return depthwise_conv1d_reference(x, weight, bias)


def keep_mask(shape, p: float, device) -> torch.Tensor:
    """0/1 keep mask with keep probability ``1 - p``: uint8 from the counter-based kernel on a GPU, float elsewhere."""
    device = torch.device(device)
    if device.type == "cuda":
        from .elementwise_hip import hip_keep_mask
        return hip_keep_mask(tuple(shape), float(p), device)
    return torch.bernoulli(torch.full(tuple(shape), 1.0 - p, device=device, dtype=torch.float32))


def dropout(x: torch.Tensor, p: float, training: bool = True) -> torch.Tensor:
    """``F.dropout`` whose GPU path draws from the device counter (no mask tensor; backward recomputes the mask)."""
    if not training or p <= 0.0:
        return x
    if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
        from .elementwise_hip import FusedDropout
        from .rng import next_seed
        return FusedDropout.apply(x, next_seed(x.device), float(p))
    return F.dropout(x, p, True)


def relu_dropout(x: torch.Tensor, p: float, training: bool = True) -> torch.Tensor:
    """``dropout(relu(x))``; one kernel per direction on a GPU when dropout is active."""
    if training and p > 0.0 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
        from .elementwise_hip import FusedReluDropout
        from .rng import next_seed
        return FusedReluDropout.apply(x, next_seed(x.device), float(p))
    return dropout(F.relu(x), p, training)


def reglu_dropout(u: torch.Tensor, p: float, training: bool = True) -> torch.Tensor:
    """``dropout(a * relu(g))`` with ``a, g = u.chunk(2, -1)``; on a GPU in bf16 one kernel per direction."""
    h = u.shape[-1] // 2
    if u.is_cuda and u.dtype == torch.bfloat16 and h % 8 == 0 and u.shape[-1] == 2 * h and u.numel() > 0:
        from .elementwise_hip import FusedReGLUDropout
        seed = None
        if training and p > 0.0:
            from .rng import next_seed
            seed = next_seed(u.device)
        return FusedReGLUDropout.apply(u, seed, float(p) if seed is not None else 0.0)
    a, g = u.chunk(2, dim=-1)
    return dropout(a * F.relu(g), p, training)
