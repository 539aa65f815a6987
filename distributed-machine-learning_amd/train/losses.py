"""Regression losses and metrics.

The reference selects its criterion from ``{mse: MSELoss, mae: L1Loss, huber: SmoothL1Loss,
mape: mape_loss}`` (/root/reference/ray-tune-hpo-regression.py:313-319) with
``mape_loss = mean(|t - o| / (t + 1e-8)) * 100`` (:245-247).  The MAPE here divides by
``|t| + eps`` so that negative targets do not flip the sign of the loss.  The integer codes are
the ones the fused HIP loss kernel switches on (``ops/csrc/head_loss.hip``).
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

LOSS_CODES: Dict[str, int] = {"mse": 0, "mae": 1, "huber": 2, "mape": 3}
MAPE_EPS = 1e-8


def mape_loss(output: torch.Tensor, target: torch.Tensor, eps: float = MAPE_EPS) -> torch.Tensor:
    return torch.mean(torch.abs(target - output) / (torch.abs(target) + eps)) * 100.0


def elementwise_loss(name: str, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    d = output - target
    if name == "mse":
        return d * d
    if name == "mae":
        return d.abs()
    if name == "huber":           # SmoothL1 with beta = 1
        a = d.abs()
        return torch.where(a < 1.0, 0.5 * d * d, a - 0.5)
    if name == "mape":
        return 100.0 * d.abs() / (target.abs() + MAPE_EPS)
    raise ValueError(f"unknown loss {name!r}")


def elementwise_loss_grad(name: str, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """d loss_i / d output_i (before the 1/batch of the mean reduction)."""
    d = output - target
    if name == "mse":
        return 2.0 * d
    if name == "mae":
        return torch.sign(d)
    if name == "huber":
        return torch.where(d.abs() < 1.0, d, torch.sign(d))
    if name == "mape":
        return 100.0 * torch.sign(d) / (target.abs() + MAPE_EPS)
    raise ValueError(f"unknown loss {name!r}")


def get_criterion(name: str) -> Callable[[torch.Tensor, torch.Tensor], torch.Tensor]:
    name = name.lower()
    if name == "mse":
        return torch.nn.MSELoss()
    if name == "mae":
        return torch.nn.L1Loss()
    if name == "huber":
        return torch.nn.SmoothL1Loss()
    if name == "mape":
        return mape_loss
    raise ValueError(f"unknown loss {name!r}")
