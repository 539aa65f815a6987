"""MLP regressor: the single-trial module, its config schema and deterministic initialisation.

The reference has no stand-alone MLP; the closest piece is the ReLU regression head of its
Transformer (``d -> 128 -> 64 -> 32 -> 16 -> 1``,
/root/reference/ray-tune-hpo-regression.py:217-221,236-240).  ``MLPRegressor`` is that shape
generalised: ``in_features -> hidden_sizes... -> 1`` with ReLU between layers.  It doubles as

* the *unpacked* view of one slot of a packed multi-trial model -- checkpoints written by the
  packed engines load into it with ``load_state_dict`` -- and
* the oracle the packed engines are tested against.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

OPTIMIZERS = ("adam", "adamw", "sgd", "rmsprop")
LOSSES = ("mse", "mae", "huber", "mape")


@dataclass
class MLPTrialSpec:
    """Normalised per-trial hyper-parameters (defaults fill what a config omits)."""
    in_features: int
    hidden_sizes: List[int]
    batch_size: int = 256
    learning_rate: float = 1e-3
    weight_decay: float = 0.0
    optimizer: str = "adamw"
    loss_function: str = "mse"
    momentum: float = 0.9
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    gradient_clipping: float = 0.0
    warmup_steps: int = 0
    total_steps: int = 0

    @classmethod
    def from_config(cls, config: Dict, in_features: int) -> "MLPTrialSpec":
        if "hidden_sizes" in config:
            hidden = [int(h) for h in config["hidden_sizes"]]
        else:
            width = int(config.get("width", config.get("d_model", 64)))
            depth = int(config.get("num_layers", config.get("depth", 2)))
            hidden = [int(config.get(f"width_{i}", width)) for i in range(depth)]
        if not hidden or min(hidden) <= 0:
            raise ValueError(f"bad hidden sizes {hidden}")
        opt = str(config.get("optimizer", "adamw")).lower()
        loss = str(config.get("loss_function", config.get("loss", "mse"))).lower()
        if opt not in OPTIMIZERS:
            raise ValueError(f"optimizer must be one of {OPTIMIZERS}, got {opt!r}")
        if loss not in LOSSES:
            raise ValueError(f"loss_function must be one of {LOSSES}, got {loss!r}")
        return cls(
            in_features=in_features, hidden_sizes=hidden,
            batch_size=int(config.get("batch_size", 256)),
            learning_rate=float(config.get("learning_rate", config.get("lr", 1e-3))),
            weight_decay=float(config.get("weight_decay", 0.0)),
            optimizer=opt, loss_function=loss,
            momentum=float(config.get("momentum", 0.9 if opt == "sgd" else 0.0)),
            beta1=float(config.get("beta1", 0.9)), beta2=float(config.get("beta2", 0.999)),
            eps=float(config.get("eps", 1e-8)),
            gradient_clipping=float(config.get("gradient_clipping", 0.0)),
            warmup_steps=int(config.get("warmup_steps", 0)), total_steps=int(config.get("total_steps", 0)),
        )

    @property
    def layer_dims(self) -> List[tuple]:
        """``[(out, in), ...]`` for the hidden layers followed by the scalar head."""
        dims, k = [], self.in_features
        for h in self.hidden_sizes:
            dims.append((h, k))
            k = h
        dims.append((1, k))
        return dims

    @property
    def num_params(self) -> int:
        return sum(o * i + o for o, i in self.layer_dims)


class MLPRegressor(nn.Module):
    def __init__(self, in_features: int, hidden_sizes: Sequence[int]):
        super().__init__()
        dims = [in_features] + list(hidden_sizes)
        self.layers = nn.ModuleList(nn.Linear(dims[i], dims[i + 1]) for i in range(len(hidden_sizes)))
        self.head = nn.Linear(dims[-1], 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for lin in self.layers:
            x = torch.relu(lin(x))
        return self.head(x).squeeze(-1)


def trial_generator(seed: int, trial_index: int, stream: int = 0) -> torch.Generator:
    """CPU generator keyed by (experiment seed, trial index, stream) -- never by pack or slot."""
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1_000_003 + int(trial_index)) * 16 + int(stream) + 0x5EED)
    return g


def init_mlp_state(spec: MLPTrialSpec, seed: int, trial_index: int) -> Dict[str, torch.Tensor]:
    """``nn.Linear``-style init (U(+-1/sqrt(fan_in)) for weight and bias) from the trial's own stream.

    Returned as an ``MLPRegressor`` ``state_dict`` on the CPU; every engine starts a trial from
    exactly these numbers, which is what makes results independent of pack, slot and GPU.
    """
    g = trial_generator(seed, trial_index)
    state: Dict[str, torch.Tensor] = {}
    dims = spec.layer_dims
    for li, (o, i) in enumerate(dims):
        bound = 1.0 / math.sqrt(i)
        # in place: the same three roundings per element as ``(u * 2 - 1) * bound`` without three temporaries
        w = torch.rand(o, i, generator=g).mul_(2).sub_(1).mul_(bound)
        b = torch.rand(o, generator=g).mul_(2).sub_(1).mul_(bound)
        name = f"layers.{li}" if li < len(dims) - 1 else "head"
        state[f"{name}.weight"], state[f"{name}.bias"] = w, b
    return state


def lr_factor(step: int, warmup_steps: int, total_steps: int) -> float:
    """Linear warm-up then linear decay to zero at ``total_steps`` (0 disables either part).

    Same shape as the reference's ``LambdaLR`` (ray-tune-hpo-regression.py:299-310), but stepped
    per optimiser step: the reference steps it per epoch against step-scale constants, which pins
    the learning rate near zero for the whole run.
    """
    if warmup_steps > 0 and step < warmup_steps:
        return (step + 1) / float(warmup_steps)
    if total_steps > 0:
        return max(0.0, (total_steps - step) / float(max(1, total_steps - warmup_steps)))
    return 1.0
