"""PackedTrainable: a model family + dataset that the framework trains as packs of trials.

Where the reference hands Ray a Python function and gets one process and one GPU per trial
(/root/reference/ray-tune-hpo-regression.py:469-475), a ``PackedTrainable`` describes *what* to
train (model family, data, epochs, metrics) and leaves *how* to the packed executor: many trials
share a GPU as slots of one multi-trial model and advance together, one grouped kernel launch per
layer per step.  The dataset is the ``with_parameters`` payload: uploaded once per GPU, shared by
every trial.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch


@dataclass
class PackedTrainable:
    model: str = "mlp"               # "mlp" | "ft_transformer" | "transformer" | "simple_transformer" | factory
    train: Tuple[torch.Tensor, torch.Tensor] = None      # (X [N, F] or sequences [N, S, F], y [N])
    val: Tuple[torch.Tensor, torch.Tensor] = None
    max_epochs: int = 16                                 # a trial that is never stopped ends here
    engine: str = "auto"                                 # "auto" | "torch" | "hip" | "graph" | "seq"
    dtype: str = "bf16"                                  # compute dtype of the HIP engine ("bf16" | "f32")
    capacity: Optional[int] = None                       # trial slots per GPU (None: HBM budget decides)
    hbm_budget_gb: float = 240.0                         # of the 288 GB, what a pack may plan for
    val_chunk: int = 4096                                # validation rows per forward launch
    checkpoint_at_end: bool = False                      # write model.pt / optimizer.pt when a trial ends
    checkpoint_freq: int = 0                             # also every k epochs (0 = never)
    on_nonfinite: str = "stop"                           # NaN/inf loss in a slot: "stop" | "error" | "ignore"
    rebalance_gap: int = 2                               # move trials between GPUs once live-trial counts differ by
                                                         # this much (0 = a trial never leaves its first GPU)
    on_rank_loss: str = "repair"                         # a peer process vanished: "repair" = the survivors re-form the
                                                         # group and restart its trials; "raise" = stop with RankLostError
    exchange: str = "lockstep"                           # ranks' metric tables: "lockstep" = gathered and decided on every
                                                         # round; "async" = the gather is posted and read one round later
                                                         # (no rank waits for the slowest within a round; decisions and
                                                         # early stops take effect one epoch late); "rungs" = posted on
                                                         # rounds where the scheduler can decide nothing, read in full
                                                         # before a round where it can (lockstep's results, fewer waits)
    name: str = "packed"
    model_kwargs: Dict[str, Any] = field(default_factory=dict)

    def __post_init__(self):
        if self.train is None or self.val is None:
            raise ValueError("PackedTrainable needs train=(X, y) and val=(X, y)")
        for X, y in (self.train, self.val):
            if X.dim() not in (2, 3) or y.dim() != 1 or X.shape[0] != y.shape[0]:
                raise ValueError("datasets must be X [N, F] (or sequences [N, S, F]) float and y [N] float")
        if self.train[0].shape[1:] != self.val[0].shape[1:]:
            raise ValueError("train/val feature shapes differ")
        if self.on_rank_loss not in ("repair", "raise"):
            raise ValueError("on_rank_loss must be 'repair' or 'raise'")
        if self.exchange not in ("lockstep", "async", "rungs"):
            raise ValueError("exchange must be 'lockstep', 'async' or 'rungs'")
        if self.model == "mlp" and self.train[0].dim() != 2:
            raise ValueError("the MLP family trains on tabular X [N, F]")

    @property
    def in_features(self) -> int:
        return int(self.train[0].shape[-1])

    @property
    def __name__(self) -> str:      # so experiment directories read "<name>_<date>"
        return self.name
