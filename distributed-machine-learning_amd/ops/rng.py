"""Device-resident random state for dropout on the GPU.

One int64 counter per device.  Every dropout site bumps it (an in-place add, stream-ordered and capturable in a
hipGraph) and takes a snapshot as its seed; the kernels of ``csrc/dropout.hip`` / ``csrc/posenc.hip`` hash
``(seed, element index)``.  Because the counter lives on the device and is advanced *by the captured work itself*,
a replayed training step draws new masks every time, and nothing depends on the framework's generator (whose
graph-registered Philox state misbehaved here once three or more captured graphs were alive).

``seed_all(value)`` pins the counter, e.g. to a function of (experiment seed, trial, epoch), which makes a trial's
masks independent of which other trials share the GPU.
"""
from __future__ import annotations

from typing import Dict

import torch

_CTR: Dict[torch.device, torch.Tensor] = {}
_STRIDE = 0x9E3779B97F4A7C15 & 0x7FFFFFFFFFFFFFFF       # odd: successive seeds differ in many bits


def counter(device) -> torch.Tensor:
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    t = _CTR.get(device)
    if t is None:
        t = torch.full((1,), 0x1234567, dtype=torch.int64, device=device)
        _CTR[device] = t
    return t


def seed_all(value: int, device="cuda") -> None:
    counter(device).fill_(int(value) & 0x7FFFFFFFFFFFFFFF)


def next_seed(device) -> torch.Tensor:
    """Advance the device counter and return a snapshot [1] int64 that stays valid for a backward pass."""
    c = counter(device)
    c.add_(_STRIDE)            # wraps modulo 2^64
    return c.clone()
