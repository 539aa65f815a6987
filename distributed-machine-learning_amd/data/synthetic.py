"""Synthetic regression datasets (no dataset can be downloaded here, and the reference's real data is private).

``create_dummy_data`` mirrors the sample script's generator
(/root/reference/ray-tune-hpo-regression-sample.py:28-55): X ~ U[0,1) of shape [n, seq, feat], first 80 % train, last
20 % validation, batch 32, shuffled training loader.  The reference draws the target independently of X
(pure noise, so no configuration can beat MSE = 1/12); ``learnable=True`` (default) makes the target a
noisy function of the inputs so that "best validation loss" measures something.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


def sequence_regression_arrays(num_samples: int = 1000, seq_len: int = 50, input_size: int = 10,
                               learnable: bool = True, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    X = rng.random((num_samples, seq_len, input_size), dtype=np.float32)
    if learnable:
        w = rng.normal(size=input_size).astype(np.float32)
        recent = X[:, -5:, :].mean(axis=1) @ w                       # depends on the last few steps
        trend = (X[:, -1, 0] - X[:, 0, 0])
        y = np.tanh(recent) + 0.5 * trend + 0.05 * rng.normal(size=num_samples).astype(np.float32)
        y = y.astype(np.float32)
    else:
        y = rng.random(num_samples, dtype=np.float32)
    return X, y


def create_dummy_data(num_samples: int = 1000, seq_len: int = 50, input_size: int = 10, batch_size: int = 32,
                      learnable: bool = True, seed: int = 0) -> Tuple[DataLoader, DataLoader]:
    X, y = sequence_regression_arrays(num_samples, seq_len, input_size, learnable, seed)
    split = int(0.8 * num_samples)
    train = TensorDataset(torch.from_numpy(X[:split]), torch.from_numpy(y[:split]))
    val = TensorDataset(torch.from_numpy(X[split:]), torch.from_numpy(y[split:]))
    g = torch.Generator().manual_seed(seed)
    return (DataLoader(train, batch_size=batch_size, shuffle=True, generator=g),
            DataLoader(val, batch_size=batch_size))


def tabular_regression(n_rows: int, n_features: int, seed: int = 0, noise: float = 0.05, device="cpu",
                       val_fraction: float = 0.2):
    """Learnable tabular target: random two-layer teacher + two pairwise interactions + noise.

    Returns ``((X_train, y_train), (X_val, y_val))`` as float32 tensors on ``device``; the target is
    standardised, so predicting the mean scores a validation MSE of 1.0.
    """
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.randn(n_rows, n_features, generator=g, device=device)
    hid = max(8, min(32, n_features))
    w1 = torch.randn(n_features, hid, generator=g, device=device) / n_features ** 0.5
    w2 = torch.randn(hid, generator=g, device=device) / hid ** 0.5
    y = torch.tanh(X @ w1) @ w2
    if n_features >= 4:
        y = y + 0.25 * X[:, 0] * X[:, 1] - 0.25 * X[:, 2] * X[:, 3]
    y = y + noise * torch.randn(n_rows, generator=g, device=device)
    y = (y - y.mean()) / y.std()
    k = int((1.0 - val_fraction) * n_rows)
    return (X[:k].contiguous(), y[:k].contiguous()), (X[k:].contiguous(), y[k:].contiguous())
