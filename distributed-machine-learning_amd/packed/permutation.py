"""Stateless per-trial shuffling: a keyed Feistel permutation of ``[0, n)``.

The reference shuffles with ``DataLoader(shuffle=True)`` on the host and copies every batch to
the GPU (/root/reference/ray-tune-hpo-regression.py:456,327).  Packed trials instead compute
"which dataset row is position ``p`` of trial ``t``'s epoch ``e``" on the fly: a 4-round Feistel
network over the smallest even-bit domain covering ``n`` with cycle walking back into ``[0, n)``.
It needs no index array (256 trials x 800k rows would be 0.8 GB of indices per epoch), gives every
(seed, trial, epoch) its own permutation, and is bit-identical between this numpy version and the
HIP gather kernel (``ops/csrc/gather.hip``) so that CPU and GPU engines see the same batches.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint32(0x9E3779B1)
_M2 = np.uint32(0x85EBCA6B)
_M3 = np.uint32(0xC2B2AE35)


def mix32(x):
    """32-bit finaliser-style hash on uint32 (scalar or array); mirrored in gather.hip."""
    x = np.asarray(x, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint32(16))) * _M2
        x = (x ^ (x >> np.uint32(13))) * _M3
        x = x ^ (x >> np.uint32(16))
    return x


def round_keys(seed: int, trial: int, epoch: int) -> np.ndarray:
    """Four 32-bit round keys for (experiment seed, trial index, epoch)."""
    with np.errstate(over="ignore"):
        base = mix32(np.uint32(seed & 0xFFFFFFFF) * _M1 + np.uint32(trial & 0xFFFFFFFF))
        base = mix32(base ^ (np.uint32(epoch & 0xFFFFFFFF) * _M1 + np.uint32(0x632BE5AB)))
        return np.array([mix32(base + np.uint32(r + 1) * _M1) for r in range(4)], dtype=np.uint32)


def half_bits(n: int) -> int:
    bits = max(2, int(n - 1).bit_length())
    return (bits + 1) // 2


def _feistel(x: np.ndarray, keys: np.ndarray, hb: int) -> np.ndarray:
    mask = np.uint32((1 << hb) - 1)
    left = (x >> np.uint32(hb)) & mask
    right = x & mask
    for r in range(4):
        f = mix32(right ^ keys[r]) & mask
        left, right = right, left ^ f
    return (left << np.uint32(hb)) | right


def permute(pos, n: int, keys: np.ndarray) -> np.ndarray:
    """Image of position(s) ``pos`` under the keyed permutation of ``[0, n)``."""
    hb = half_bits(n)
    x = _feistel(np.asarray(pos, dtype=np.uint32), keys, hb)
    out_of_range = x >= np.uint32(n)
    while out_of_range.any():            # cycle walking; terminates because _feistel is a bijection
        x = np.where(out_of_range, _feistel(x, keys, hb), x)
        out_of_range = x >= np.uint32(n)
    return x.astype(np.int64)


def epoch_permutation(n: int, seed: int, trial: int, epoch: int) -> np.ndarray:
    return permute(np.arange(n, dtype=np.uint32), n, round_keys(seed, trial, epoch))
