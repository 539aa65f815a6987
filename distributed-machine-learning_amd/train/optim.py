"""Optimiser construction and the warm-up/decay LR schedule for single-trial (unpacked) training.

The reference maps ``{"adam", "adamw", "sgd", "rmsprop"}`` to ``torch.optim`` classes and then
passes ``momentum=`` to all four (/root/reference/ray-tune-hpo-regression.py:253-258,291-296),
which raises for Adam/AdamW.  ``build_optimizer`` passes each optimiser only what it accepts.
The integer codes are shared with the fused HIP optimiser kernel (``ops/csrc/optim.hip``).
"""
from __future__ import annotations

from typing import Dict, Iterable

import torch

from ..models.mlp import lr_factor

OPT_CODES: Dict[str, int] = {"adam": 0, "adamw": 1, "sgd": 2, "rmsprop": 3}

optimizer_mapping = {
    "adam": torch.optim.Adam,
    "adamw": torch.optim.AdamW,
    "sgd": torch.optim.SGD,
    "rmsprop": torch.optim.RMSprop,
}


def build_optimizer(name: str, params: Iterable, lr: float, weight_decay: float = 0.0, momentum: float = 0.0,
                    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> torch.optim.Optimizer:
    name = name.lower()
    if name in ("adam", "adamw"):
        return optimizer_mapping[name](params, lr=lr, weight_decay=weight_decay, betas=(beta1, beta2), eps=eps)
    if name == "sgd":
        return torch.optim.SGD(params, lr=lr, weight_decay=weight_decay, momentum=momentum)
    if name == "rmsprop":
        return torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay, momentum=momentum, eps=eps, alpha=0.99)
    raise ValueError(f"unknown optimizer {name!r}")


def warmup_decay_scheduler(optimizer: torch.optim.Optimizer, warmup_steps: int, total_steps: int):
    """``LambdaLR`` with linear warm-up then linear decay; call ``.step()`` once per optimiser step."""
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: lr_factor(s, warmup_steps, total_steps))
