"""Glucose-forecasting data pipeline: ``.npy`` tables -> 81-feature windows -> train/val loaders.

Same stages as the reference (/root/reference/ray-tune-hpo-regression.py:414-459): load a
``{"columns", "data"}`` table, select the feature columns, drop duplicate columns, take
``"Historic Glucose mg/dL"`` as the label, cut non-overlapping 96-step windows, split 70/30 with a
fixed seed, wrap in loaders of batch 32.  Differences: paths are arguments (the reference hard-codes
a home directory); tables are read with ``allow_pickle=False`` first (a plain ``.npz``/structured
layout) and only fall back to pickled dicts for files the caller vouches for; each window gets ONE
label (its last row) -- the reference passes the unwindowed label vector, which only lines up when the
table already holds one row per window; and a synthetic stand-in with the same columns exists because
the real files are private.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset

from . import features as feat
from .windows import split_into_intervals, window_labels


def load_dataframe_from_npy(path: str, trusted: bool = False):
    """Load a table saved as ``np.save(path, {"columns": [...], "data": ndarray})`` or as an ``.npz``."""
    import pandas as pd
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return pd.DataFrame(z["data"], columns=[str(c) for c in z["columns"]])
    try:
        arr = np.load(path, allow_pickle=False)
    except ValueError:
        if not trusted:
            raise ValueError(f"{path} holds pickled objects; pass trusted=True only for files you wrote yourself")
        arr = np.load(path, allow_pickle=True)
    obj = arr.item() if arr.dtype == object else arr
    if isinstance(obj, dict):
        return pd.DataFrame(obj["data"], columns=list(obj["columns"]))
    if getattr(obj, "dtype", None) is not None and obj.dtype.names:
        return pd.DataFrame({n: obj[n] for n in obj.dtype.names})
    raise ValueError(f"unrecognised table layout in {path}")


def synthetic_glucose_frame(n_rows: int = 96 * 120, seed: int = 0):
    """A DataFrame with the 81 feature columns + label, statistically glucose-like (daily rhythm + meals)."""
    import pandas as pd
    rng = np.random.default_rng(seed)
    t = np.arange(n_rows)
    minute = (t * 15) % 1440
    day = (t * 15 // 1440) % 7
    base = {"HeartRate": 70 + 10 * np.sin(2 * np.pi * minute / 1440) + rng.normal(0, 4, n_rows),
            "Sleep": ((minute < 420) | (minute > 1380)).astype(np.float64),
            "Intensity": np.clip(rng.gamma(1.5, 8, n_rows) * (minute > 420), 0, None),
            "Steps": np.clip(rng.gamma(1.2, 60, n_rows) * (minute > 420), 0, None)}
    cols = dict(base)
    for sig, series in base.items():
        s = pd.Series(series)
        unit = "" if sig == "HeartRate" else "min"
        for w in feat.ROLLING_WINDOWS_MIN:
            k = max(1, w // 15)
            r = s.rolling(k, min_periods=1)
            cols[f"{sig}_{w}{unit}_Mean"] = r.mean().to_numpy()
            cols[f"{sig}_{w}{unit}_Std"] = r.std().fillna(0.0).to_numpy()
    cols["MinuteOfDay_Sin"], cols["MinuteOfDay_Cos"] = np.sin(2 * np.pi * minute / 1440), np.cos(2 * np.pi * minute / 1440)
    cols["DayOfWeek_Sin"], cols["DayOfWeek_Cos"] = np.sin(2 * np.pi * day / 7), np.cos(2 * np.pi * day / 7)
    cols["Is_Weekend"] = (day >= 5).astype(np.float64)
    glucose = (100 + 15 * np.sin(2 * np.pi * (minute - 480) / 1440) + 0.3 * (cols["HeartRate_60_Mean"] - 70)
               - 0.02 * cols["Steps_60min_Mean"] + 8 * cols["Sleep_180min_Mean"] + rng.normal(0, 3, n_rows))
    cols[feat.LABEL_COLUMN] = glucose
    return pd.DataFrame(cols)


def frames_to_loaders(frame, feature_columns: Optional[Sequence[str]] = None, interval_size: int = 96,
                      stride: int = 96, test_size: float = 0.3, random_state: int = 42, batch_size: int = 32,
                      standardize: bool = True) -> Tuple[DataLoader, DataLoader]:
    from sklearn.model_selection import train_test_split
    cols = list(feature_columns or feat.features)
    X = frame[cols]
    X = X.loc[:, ~X.columns.duplicated()]
    y = frame[feat.LABEL_COLUMN].to_numpy(dtype=np.float32)
    Xw = np.ascontiguousarray(split_into_intervals(X.to_numpy(dtype=np.float32), interval_size, stride))
    yw = np.ascontiguousarray(window_labels(y, interval_size, stride, where="last"))
    Xtr, Xva, ytr, yva = train_test_split(Xw, yw, test_size=test_size, random_state=random_state)
    if standardize:            # statistics from the training windows only
        mu = Xtr.reshape(-1, Xtr.shape[-1]).mean(0)
        sd = Xtr.reshape(-1, Xtr.shape[-1]).std(0) + 1e-6
        Xtr, Xva = (Xtr - mu) / sd, (Xva - mu) / sd
    g = torch.Generator().manual_seed(random_state)
    tr = TensorDataset(torch.from_numpy(Xtr.astype(np.float32)), torch.from_numpy(ytr))
    va = TensorDataset(torch.from_numpy(Xva.astype(np.float32)), torch.from_numpy(yva))
    return DataLoader(tr, batch_size=batch_size, shuffle=True, generator=g), DataLoader(va, batch_size=batch_size)


def get_data_loaders(id_: str, data_dir: Optional[str] = None, trusted: bool = False, **kw) -> Tuple[DataLoader, DataLoader]:
    """Loaders for subject ``id_``: reads ``<data_dir>/<id>_combined_data.npy`` (+ ``_main_data.npy``) when present,
    otherwise builds the synthetic stand-in seeded by the subject id."""
    if data_dir:
        combined = os.path.join(data_dir, f"{id_}_combined_data.npy")
        if os.path.exists(combined):
            frame = load_dataframe_from_npy(combined, trusted=trusted)
            main = os.path.join(data_dir, f"{id_}_main_data.npy")
            if feat.LABEL_COLUMN not in frame.columns and os.path.exists(main):
                frame[feat.LABEL_COLUMN] = load_dataframe_from_npy(main, trusted=trusted)[feat.LABEL_COLUMN].to_numpy()
            return frames_to_loaders(frame, **kw)
    seed = sum(ord(c) for c in str(id_))
    return frames_to_loaders(synthetic_glucose_frame(seed=seed), **kw)
