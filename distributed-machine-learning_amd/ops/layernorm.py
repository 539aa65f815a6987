"""Fused dropout + residual add + LayerNorm: ``LN(residual + dropout(x))``.

This is the step that closes both halves of a post-norm encoder block
(/root/reference/ray-tune-hpo-regression.py:172,176).  As separate PyTorch ops it is three
memory-bound passes over [tokens, d]; fused it is one read of each input and one write.  On a GPU
``ops/csrc/layernorm.hip`` runs it with one wave per row (forward and backward); elsewhere the
reference below is used.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def residual_layer_norm_reference(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                                  bias: torch.Tensor, eps: float = 1e-5,
                                  keep_mask: Optional[torch.Tensor] = None, dropout_p: float = 0.0) -> torch.Tensor:
    if keep_mask is not None:
        x = x * keep_mask.to(x.dtype) * (1.0 / (1.0 - dropout_p))
    return F.layer_norm(residual + x, (x.shape[-1],), weight, bias, eps)


def residual_layer_norm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                        eps: float = 1e-5, dropout_p: float = 0.0) -> torch.Tensor:
    fused = x.is_cuda and x.dtype == torch.float32 and x.shape[-1] <= 2048 and x.shape[-1] % 4 == 0
    if fused:
        from .layernorm_hip import FusedResidualLayerNorm, available
        if available():
            seed = None
            if dropout_p > 0.0:           # the mask is drawn inside the kernel from a device-resident seed
                from .rng import next_seed
                seed = next_seed(x.device)
            return FusedResidualLayerNorm.apply(x, residual, weight, bias, eps, None, dropout_p, seed)
    keep = None
    if dropout_p > 0.0:
        from .elementwise import keep_mask
        keep = keep_mask(x.shape, dropout_p, x.device)
    return residual_layer_norm_reference(x, residual, weight, bias, eps, keep, dropout_p)


def residual_layer_norm_mixed(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                              eps: float = 1e-5, dropout_p: float = 0.0, want_bf16: bool = True):
    """``(y, y16)``: ``y = LayerNorm(residual + dropout(x))`` in f32 (the residual stream) and its bf16 copy for the next
    GEMM (``None`` when ``want_bf16`` is false).  ``x`` may be the bf16 output of an autocast GEMM: on a GPU with rows of at
    most 256 columns one kernel per direction does all of it (``csrc/layernorm.hip``, mixed forms); elsewhere the f32 path
    plus explicit casts."""
    d = x.shape[-1]
    if (x.is_cuda and d <= 256 and d % 4 == 0 and x.dtype in (torch.float32, torch.bfloat16)
            and residual.dtype == torch.float32):
        from .layernorm_hip import MixedResidualLayerNorm, available
        if available():
            seed = None
            if dropout_p > 0.0:
                from .rng import next_seed
                seed = next_seed(x.device)
            out = MixedResidualLayerNorm.apply(x, residual, weight, bias, eps, dropout_p, seed, bool(want_bf16))
            return out if want_bf16 else (out, None)
    y = residual_layer_norm(x.float(), residual.float(), weight, bias, eps, dropout_p)
    return y, (y.to(torch.bfloat16) if want_bf16 else None)
