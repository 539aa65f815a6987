"""Per-trial checkpoints: ``checkpoint_<iter:06d>/{model.pt, optimizer.pt, meta.json}``.

The reference never saves a model (its trainables end at ``tune.report``,
/root/reference/ray-tune-hpo-regression.py:373); the layout here is the "per-trial model
state_dict plus result JSON" contract: ``model.pt`` is a plain ``torch.save``d ``state_dict``
whose keys are those of an ordinary single-trial ``nn.Module``, so it loads with
``torch.load(..., weights_only=True)`` outside this framework even when the trial was trained as
one slot of a packed multi-trial model.
"""
from __future__ import annotations

import json
import os
import pickle
import re
import shutil
from typing import Any, Dict, List, Optional

from .logger import dump_json


def checkpoint_dir(trial_dir: str, iteration: int) -> str:
    return os.path.join(trial_dir, f"checkpoint_{int(iteration):06d}")


def list_checkpoints(trial_dir: str) -> List[str]:
    if not trial_dir or not os.path.isdir(trial_dir):
        return []
    names = sorted(n for n in os.listdir(trial_dir) if re.fullmatch(r"checkpoint_\d{6}", n))
    return [os.path.join(trial_dir, n) for n in names]


def save_checkpoint(trial_dir: str, iteration: int, checkpoint: Any, keep_num: Optional[int] = None) -> str:
    """Persist ``checkpoint`` (a dict) under ``trial_dir``.

    Keys ``model`` / ``optimizer`` holding tensors dicts are written with ``torch.save`` as
    ``model.pt`` / ``optimizer.pt``; everything JSON-able goes to ``meta.json``; anything else is
    pickled to ``extra.pkl``.
    """
    import torch
    d = checkpoint_dir(trial_dir, iteration)
    os.makedirs(d, exist_ok=True)
    meta: Dict[str, Any] = {"training_iteration": int(iteration)}
    extra: Dict[str, Any] = {}
    if not isinstance(checkpoint, dict):
        checkpoint = {"value": checkpoint}
    for k, v in checkpoint.items():
        if k in ("model", "optimizer") and isinstance(v, dict):
            torch.save(v, os.path.join(d, f"{k}.pt"))
        else:
            try:
                json.dumps(v)
                meta[k] = v
            except (TypeError, ValueError):
                extra[k] = v
    dump_json(meta, os.path.join(d, "meta.json"))
    if extra:
        with open(os.path.join(d, "extra.pkl"), "wb") as f:
            pickle.dump(extra, f)
    if keep_num:
        for old in list_checkpoints(trial_dir)[:-keep_num]:
            shutil.rmtree(old, ignore_errors=True)
    return d


def load_checkpoint(path: str, map_location: Any = "cpu") -> Dict:
    """Inverse of ``save_checkpoint`` (``extra.pkl`` is only ever a file this framework wrote)."""
    import torch
    out: Dict[str, Any] = {}
    meta_path = os.path.join(path, "meta.json")
    if os.path.exists(meta_path):
        with open(meta_path) as f:
            out.update(json.load(f))
    for k in ("model", "optimizer"):
        p = os.path.join(path, f"{k}.pt")
        if os.path.exists(p):
            out[k] = torch.load(p, map_location=map_location, weights_only=True)
    p = os.path.join(path, "extra.pkl")
    if os.path.exists(p):
        with open(p, "rb") as f:
            out.update(pickle.load(f))
    return out


def latest_checkpoint(trial_dir: str) -> Optional[str]:
    cps = list_checkpoints(trial_dir)
    return cps[-1] if cps else None
