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
