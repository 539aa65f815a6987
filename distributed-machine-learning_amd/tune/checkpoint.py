"""Per-trial checkpoints: ``checkpoint_<iter:06d>/{model.pt, optimizer.pt, <other tensor entries>.pt, meta.json}``.

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


def _tensor_tree(v: Any) -> bool:
    """True for what ``torch.load(..., weights_only=True)`` reads back: tensors, numbers, strings, None and lists /
    tuples / string-keyed dicts of them."""
    import torch
    if isinstance(v, torch.Tensor) or v is None or isinstance(v, (bool, int, float, str)):
        return True
    if isinstance(v, dict):
        return all(isinstance(k, (str, int)) and _tensor_tree(x) for k, x in v.items())
    if isinstance(v, (list, tuple)):
        return all(_tensor_tree(x) for x in v)
    return False


def _plain(v: Any) -> Any:
    """numpy arrays and scalars become tensors / Python numbers, containers are walked; everything else is unchanged."""
    import numpy as np
    import torch
    if isinstance(v, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(v))
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(x) for x in v)
    return v


def save_checkpoint(trial_dir: str, iteration: int, checkpoint: Any, keep_num: Optional[int] = None) -> str:
    """Persist ``checkpoint`` (a dict) under ``trial_dir``.

    Everything JSON-able goes to ``meta.json``; every other entry must be a tree of tensors / numpy arrays / numbers /
    strings and is written with ``torch.save`` as ``<key>.pt`` (``model.pt`` and ``optimizer.pt`` among them), a file
    that ``torch.load(..., weights_only=True)`` reads without executing anything from it.  A value that is neither (an
    arbitrary Python object) raises ``TypeError``: a checkpoint directory is something users copy between machines, so
    nothing in it is ever unpickled.
    """
    import torch
    d = checkpoint_dir(trial_dir, iteration)
    os.makedirs(d, exist_ok=True)
    meta: Dict[str, Any] = {"training_iteration": int(iteration)}
    tensor_keys: List[str] = []
    if not isinstance(checkpoint, dict):
        checkpoint = {"value": checkpoint}
    for k, v in checkpoint.items():
        if k not in ("model", "optimizer"):
            try:
                json.dumps(v)
                meta[k] = v
                continue
            except (TypeError, ValueError):
                pass
        v = _plain(v)
        if not re.fullmatch(r"[A-Za-z0-9_\-]+", str(k)) or not _tensor_tree(v):
            raise TypeError(f"checkpoint entry {k!r} ({type(v).__name__}) is neither JSON-able nor a tree of tensors; "
                            "convert it before reporting the checkpoint")
        torch.save(v, os.path.join(d, f"{k}.pt"))
        tensor_keys.append(str(k))
    meta["tensor_entries"] = tensor_keys
    dump_json(meta, os.path.join(d, "meta.json"))
    if keep_num:
        for old in list_checkpoints(trial_dir)[:-keep_num]:
            shutil.rmtree(old, ignore_errors=True)
    return d


def load_checkpoint(path: str, map_location: Any = "cpu") -> Dict:
    """Inverse of ``save_checkpoint``: ``meta.json`` plus every ``<key>.pt`` it lists, each read with
    ``weights_only=True``."""
    import torch
    out: Dict[str, Any] = {}
    keys = ["model", "optimizer"]
    meta_path = os.path.join(path, "meta.json")
    if os.path.exists(meta_path):
        with open(meta_path) as f:
            meta = json.load(f)
        keys = list(dict.fromkeys(keys + list(meta.pop("tensor_entries", []))))
        out.update(meta)
    for k in keys:
        p = os.path.join(path, f"{k}.pt")
        if os.path.exists(p):
            out[k] = torch.load(p, map_location=map_location, weights_only=True)
    return out


def latest_checkpoint(trial_dir: str) -> Optional[str]:
    cps = list_checkpoints(trial_dir)
    return cps[-1] if cps else None
