"""Host-side planning of the row-block training step (ops/csrc/rowblock.hip): which trials' 16-row blocks are shared by
several workgroups, and the order of the work list the kernel's workgroups index with ``blockIdx.x``.

Pure functions of small tuples, so that what the kernel assumes about its work list -- every block listed once per part,
the parts of a block 8 entries apart, widths divisible by the parts -- is checked by CPU tests
(tests/test_layout_properties.py) and not only by GPU runs.  The step these lists drive replaces the per-layer launches
of /root/reference/ray-tune-hpo-regression-sample.py:110-120 for packs of a few surviving trials.
"""
from __future__ import annotations

from typing import Dict, List, NamedTuple, Sequence, Tuple

import numpy as np

N_XCD = 8                 # workgroup b is dispatched to XCD b % 8
CUS_PER_XCD = 32          # a row-block workgroup fills one CU (LDS)
ROW0_BITS, PART_SHIFT, NPARTS_SHIFT = 20, 20, 24


class RbTrial(NamedTuple):
    """What the planner needs to know about one trial of a chain."""
    slot: int
    n_blocks: int                 # 16-row blocks of its mini-batch
    hidden: Tuple[int, ...]       # layer widths
    in_dims: Tuple[int, ...]      # contraction length of every layer

    @property
    def macs(self) -> int:
        """Multiply-adds per row of one block's chain: every forward layer and every input gradient but layer 0's."""
        fwd = sum(h * k for h, k in zip(self.hidden, self.in_dims))
        return fwd + sum(h * k for h, k in zip(self.hidden[1:], self.in_dims[1:]))

    @property
    def boundaries(self) -> int:
        """Exchanges a split block makes: every forward layer's output and all but the last two input gradients travel."""
        nl = len(self.hidden)
        return nl + max(nl - 2, 0)


def can_split(trial: RbTrial, n: int) -> bool:
    """Every layer must give each part a whole number of 16-column tiles."""
    return all(h % (16 * n) == 0 for h in trial.hidden)


def split_parts(trials: Sequence[RbTrial], most: int, exchange_cost: int, budget: int) -> Dict[int, int]:
    """Workgroups per 16-row block for every trial of a chain (1, 2 or 4).  A block's chain of layers runs on one CU, and
    a step is as long as the longest chain; splitting a block over n workgroups cuts its weight stream and matrix
    instructions by n at the price of one exchange through memory per layer boundary (``exchange_cost`` multiply-adds
    each).  Greedy: the trial with the longest chain is split further while that shortens it and the work list stays
    within ``budget`` workgroups (one per CU: the parts of a block wait for each other, so all of them should be running)."""
    parts = {t.slot: 1 for t in trials}
    if most <= 1 or not trials:
        return parts
    by_slot = {t.slot: t for t in trials}

    def cost(t: RbTrial, n: int) -> float:
        return t.macs / n + (t.boundaries * exchange_cost if n > 1 else 0)

    total = sum(t.n_blocks for t in trials)
    while True:
        t = max(trials, key=lambda t: (cost(t, parts[t.slot]), -t.slot))
        n = parts[t.slot] * 2
        if n > most or not can_split(t, n) or cost(t, n) >= cost(t, parts[t.slot]):
            break
        if total + t.n_blocks * (n - parts[t.slot]) > budget:
            break
        total += t.n_blocks * (n - parts[t.slot])
        parts[t.slot] = n
    assert all(can_split(by_slot[s], n) for s, n in parts.items())
    return parts


def work_list(trials: Sequence[RbTrial], parts: Dict[int, int], rows: int = 16) -> List[Tuple[int, int, int, int]]:
    """The work list as (slot, first row, part, parts) per workgroup, slot -1 for padding entries; its length is a
    multiple of 8.  Trials are dealt to the XCDs heaviest first, each to the XCD with the least work so far whose first
    round (32 CUs) still has room -- the blocks of a trial stream the same weights, so they stay on one XCD's L2 as far
    as they fit and spill onto the next emptiest XCD rather than queue behind each other.  An XCD's list starts with its
    heaviest blocks, so that the long chains start first and the short ones fill in behind them.  The parts of a split
    block follow each other in their XCD's list: 8 entries apart in the whole list, consecutive in that XCD's dispatch
    order, which is what lets a part wait for its partners without ever waiting for a workgroup that cannot start."""
    per_xcd: List[List[Tuple[float, int, int, int]]] = [[] for _ in range(N_XCD)]
    load = [0.0] * N_XCD
    for t in sorted(trials, key=lambda t: (-t.macs / parts[t.slot], t.slot)):
        n, macs = parts[t.slot], t.macs / parts[t.slot]
        groups = list(range(t.n_blocks))
        while groups:
            open_ = [k for k in range(N_XCD) if CUS_PER_XCD - len(per_xcd[k]) >= n]
            if open_:
                x = min(open_, key=lambda k: (load[k], k))
                take = groups[:(CUS_PER_XCD - len(per_xcd[x])) // n]
            else:                                    # every XCD's first round is full: the rest queues on the shortest list
                x = min(range(N_XCD), key=lambda k: (len(per_xcd[k]), load[k], k))
                take = groups[:max(1, CUS_PER_XCD // n)]
            groups = groups[len(take):]
            per_xcd[x].extend((macs, t.slot, r * rows, part) for r in take for part in range(n))
            load[x] += macs * len(take) * n
    depth = max((len(v) for v in per_xcd), default=0)
    out = [(-1, 0, 0, 1)] * (depth * N_XCD)
    for x in range(N_XCD):
        for k, (_, slot, row0, part) in enumerate(sorted(per_xcd[x], key=lambda e: (-e[0], e[1], e[2], e[3]))):
            out[k * N_XCD + x] = (slot, row0, part, parts[slot])
    return out


def encode(entries: Sequence[Tuple[int, int, int, int]], dtype) -> np.ndarray:
    """Entries as the kernel reads them (``MtRbBlock``): slot, and the first row with the part index in bits 20-23 and the
    number of parts in bits 24-27 (0 for an unsplit block)."""
    arr = np.zeros(len(entries), dtype=dtype)
    for i, (slot, row0, part, n) in enumerate(entries):
        assert 0 <= row0 < (1 << ROW0_BITS) and 0 <= part < n <= 4
        arr[i] = (slot, row0 | (part << PART_SHIFT) | ((n if n > 1 else 0) << NPARTS_SHIFT))
    return arr
