"""Seeding helpers: one experiment seed fans out into independent, placement-free streams."""
from __future__ import annotations

import random

import numpy as np
import torch


def seed_everything(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)


def derive_seed(seed: int, *keys: int) -> int:
    """Deterministic 63-bit seed for (experiment seed, key...), e.g. (seed, trial_index, stream)."""
    x = (seed & 0xFFFFFFFFFFFFFFFF) ^ 0x9E3779B97F4A7C15
    for k in keys:
        x = (x ^ (k & 0xFFFFFFFFFFFFFFFF)) * 0xBF58476D1CE4E5B9 & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
    return x & 0x7FFFFFFFFFFFFFFF
