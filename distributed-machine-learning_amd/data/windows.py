"""Sliding-window segmentation of time series.

``split_into_intervals`` has the contract of the reference helper of the same name
(/root/reference/ray-tune-hpo-regression.py:403-411): windows of ``interval_size`` rows every ``stride`` rows,
``(N - size) // stride + 1`` of them, stacked on a new leading axis.  It is a zero-copy strided view
instead of a Python loop, so segmenting a 100M-row table costs nothing.
"""
from __future__ import annotations

import numpy as np


def split_into_intervals(data, interval_size: int, stride: int) -> np.ndarray:
    arr = np.asarray(data)
    if interval_size <= 0 or stride <= 0:
        raise ValueError("interval_size and stride must be positive")
    n = arr.shape[0]
    if n < interval_size:
        return np.empty((0, interval_size) + arr.shape[1:], dtype=arr.dtype)
    count = (n - interval_size) // stride + 1
    view = np.lib.stride_tricks.sliding_window_view(arr, interval_size, axis=0)[::stride][:count]
    # sliding_window_view puts the window axis last: [count, ..., size] -> [count, size, ...]
    return np.moveaxis(view, -1, 1)


def window_labels(labels, interval_size: int, stride: int, where: str = "last") -> np.ndarray:
    """One target per window: the label at the window's last row (forecast) or the window mean."""
    w = split_into_intervals(np.asarray(labels), interval_size, stride)
    return w[:, -1] if where == "last" else w.mean(axis=1)
