"""Autograd binding of the fused short-sequence attention kernels (``csrc/attention.hip``)."""
from __future__ import annotations

import torch


def available() -> bool:
    return False
