"""HBM budgeting: how many trial slots fit on one MI355X.

A pack's footprint is linear in its slot count: per slot, five parameter-sized slabs (f32 master, f32
moments x2, bf16 shadow, and ``ksplits`` f32 gradient slabs) plus, per hidden layer, one bf16 activation
and one bf16 activation-gradient buffer of ``rows x width``.  With 288 GB per GPU the answer is almost
always "more slots than trials"; the budget exists so that a search over large widths degrades into
several passes over the pack instead of an out-of-memory kill.
"""
from __future__ import annotations

from dataclasses import dataclass

MI355X_HBM_BYTES = 288 * 10 ** 9


@dataclass
class PackFootprint:
    per_slot_bytes: int
    fixed_bytes: int

    def slots_for(self, budget_bytes: float) -> int:
        return max(0, int((budget_bytes - self.fixed_bytes) // max(self.per_slot_bytes, 1)))

    def total(self, slots: int) -> int:
        return self.fixed_bytes + slots * self.per_slot_bytes


def mlp_pack_footprint(in_features: int, max_width: int, max_layers: int, max_rows: int, n_train: int, n_val: int,
                       gradient_slabs: int = 4, dtype_bytes: int = 2) -> PackFootprint:
    """``dtype_bytes``: 2 for bf16 packs (activations bf16, plus a bf16 parameter shadow), 4 for exact-f32 packs (f32
    activations, no shadow)."""
    fpad = (in_features + 7) // 8 * 8
    wp = (max_width + 7) // 8 * 8
    params = max_width * fpad + max_width + (max_layers - 1) * (max_width * wp + max_width) + wp + 8
    params = (params + 63) // 64 * 64
    shadow = 2 if dtype_bytes == 2 else 0
    slabs = params * (4 + 4 + 4 + shadow + 4 * gradient_slabs)       # P, M, V, Pb, G[splits]
    acts = max_rows * (fpad * dtype_bytes + 4) + 2 * max_layers * max_rows * wp * dtype_bytes
    scratch = (max_rows // 32 + 1) * (wp + 8) * 4 + ((params + 2047) // 2048) * 4
    fixed = n_train * (in_features * 4 + 4) + n_val * (fpad * dtype_bytes + 4)
    return PackFootprint(per_slot_bytes=slabs + acts + scratch, fixed_bytes=fixed)


def default_capacity(in_features: int, max_width: int, max_layers: int, max_rows: int, n_train: int, n_val: int,
                     budget_gb: float = 240.0, hard_cap: int = 1024, dtype_bytes: int = 2) -> int:
    """Slots the budget allows, at most ``hard_cap``: the engine allocates every slot's slabs up front, and beyond about a
    thousand resident trials per GPU the late rounds of a search are amortised already."""
    fp = mlp_pack_footprint(in_features, max_width, max_layers, max_rows, n_train, n_val, dtype_bytes=dtype_bytes)
    return max(1, min(hard_cap, fp.slots_for(budget_gb * 10 ** 9)))
