"""Property tests of the pure-Python index arithmetic that ends up as raw device pointers: slot layouts and tile lists."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from mitune.models.mlp import MLPTrialSpec
from mitune.packed.engine_hip import N_XCD, SlotLayout, balanced_tiles, ceil8


@settings(max_examples=200, deadline=None)
@given(f=st.integers(1, 300), hidden=st.lists(st.integers(1, 700), min_size=1, max_size=6))
def test_slot_layout_regions_are_aligned_disjoint_and_cover_every_parameter(f, hidden):
    fpad = ceil8(f)
    lay = SlotLayout(MLPTrialSpec(f, hidden), fpad)
    regions = []
    for key, off, (rows, cols), ld in lay.names():
        assert off % 8 == 0, key                               # every tensor starts on a 16-byte (bf16) / 32-byte (f32) line
        assert ld >= cols
        if key.endswith("weight") and key != "head.weight":
            assert ld % 8 == 0                                 # operand rows start 16-byte aligned
        span = (rows - 1) * ld + cols
        regions.append((off, off + span, key))
    regions.sort()
    for (a0, a1, ka), (b0, b1, kb) in zip(regions, regions[1:]):
        assert a1 <= b0, (ka, kb)                              # no overlap
    assert regions[-1][1] <= lay.total
    # the state_dict shapes are the plain module's
    shapes = {k: s for k, _, s, _ in lay.names()}
    dims = [f] + list(hidden)
    for l, h in enumerate(hidden):
        assert shapes[f"layers.{l}.weight"] == (h, dims[l]) and shapes[f"layers.{l}.bias"] == (1, h)
    assert shapes["head.weight"] == (1, hidden[-1])
    # layer l's row stride is what layer l-1 writes (its padded width), layer 0's is the padded feature count
    assert lay.in_ld[0] == fpad and lay.in_ld[1:] == [ceil8(h) for h in hidden[:-1]]


groups_st = st.lists(st.tuples(st.integers(1, 9), st.integers(1, 9), st.integers(0, 12)), min_size=0, max_size=40)


@settings(max_examples=200, deadline=None)
@given(shapes=groups_st, ksplits=st.integers(1, 5))
def test_balanced_tiles_lists_every_tile_exactly_once_on_its_groups_xcd(shapes, ksplits):
    groups = [(gi * 3 + 1, ti, tj, w) for gi, (ti, tj, w) in enumerate(shapes)]      # non-contiguous group numbers
    tiles = balanced_tiles(groups, ksplits)
    assert len(tiles) % N_XCD == 0
    live = tiles[tiles["group"] >= 0]
    pad = tiles[tiles["group"] < 0]
    assert np.all(pad["group"] == -1)                                                # padding entries are group = -1
    want = {(g, ti | (sp << 8), tj) for g, tn, tjn, _ in groups for ti in range(tn) for tj in range(tjn) for sp in range(ksplits)}
    got = [(int(t["group"]), int(t["ti"]), int(t["tj"])) for t in live]
    assert len(got) == len(set(got)) == len(want) and set(got) == want               # every tile exactly once
    # workgroup ids are dealt round-robin to the 8 XCDs: all tiles of one group carry the same id % 8
    where = {}
    for idx, t in enumerate(tiles):
        if t["group"] >= 0:
            where.setdefault(int(t["group"]), set()).add(idx % N_XCD)
    assert all(len(v) == 1 for v in where.values())
    # no longer than the fullest bucket needs
    if len(live):
        per = {}
        for g, tn, tjn, _ in groups:
            per[next(iter(where[g]))] = per.get(next(iter(where[g])), 0) + tn * tjn * ksplits
        assert len(tiles) == N_XCD * max(per.values())
    else:
        assert len(tiles) == 0


def test_tile_split_index_fits_its_field():
    t = balanced_tiles([(0, 200, 1, 1)], ksplits=3)
    live = t[t["group"] >= 0]
    assert set(int(x) & 0xFF for x in live["ti"]) == set(range(200)) and set(int(x) >> 8 for x in live["ti"]) == {0, 1, 2}


# ---- row-block work lists (packed/rowblock_plan.py): what the kernel assumes about the list it indexes with blockIdx.x
from mitune.packed import rowblock_plan as rp  # noqa: E402

rb_trials_st = st.lists(
    st.tuples(st.sampled_from([1, 2, 4, 8, 13, 16, 32, 64]),
              st.lists(st.sampled_from([32, 64, 96, 128, 160, 224, 256, 384, 512]), min_size=1, max_size=5)),
    min_size=1, max_size=20)


def _rb_trials(raw, features=64):
    return [rp.RbTrial(slot, nblk, tuple(hidden), tuple([features] + hidden[:-1])) for slot, (nblk, hidden) in enumerate(raw)]


@settings(max_examples=200, deadline=None)
@given(raw=rb_trials_st, most=st.sampled_from([1, 2, 4]), cost=st.sampled_from([0, 20000, 50000, 400000]),
       budget=st.sampled_from([16, 100, 256, 400]))
def test_row_block_split_rule_only_splits_what_divides_and_stays_within_its_budget(raw, most, cost, budget):
    trials = _rb_trials(raw)
    parts = rp.split_parts(trials, most, cost, budget)
    assert set(parts) == {t.slot for t in trials}
    unsplit = sum(t.n_blocks for t in trials)
    for t in trials:
        n = parts[t.slot]
        assert n in (1, 2, 4) and n <= most
        assert all(h % (16 * n) == 0 for h in t.hidden)         # every part gets whole 16-column tiles of every layer
    total = sum(t.n_blocks * parts[t.slot] for t in trials)
    assert total <= max(budget, unsplit)                          # splits never push the list past the budget
    if cost >= 400000:                                            # an exchange dearer than any chain: nothing is split
        assert all(n == 1 for n in parts.values())


@settings(max_examples=200, deadline=None)
@given(raw=rb_trials_st, most=st.sampled_from([1, 2, 4]), budget=st.sampled_from([64, 256, 400]))
def test_row_block_work_list_lists_every_block_once_per_part_with_partners_eight_entries_apart(raw, most, budget):
    trials = _rb_trials(raw)
    parts = rp.split_parts(trials, most, 0, budget)
    wl = rp.work_list(trials, parts, rows=16)
    assert len(wl) % rp.N_XCD == 0
    seen = {}
    for i, (slot, row0, part, n) in enumerate(wl):
        if slot < 0:
            continue
        assert n == parts[slot] and 0 <= part < n and row0 % 16 == 0
        assert (slot, row0, part) not in seen
        seen[(slot, row0, part)] = i
    for t in trials:                                              # complete: every block, every part
        for b in range(t.n_blocks):
            for part in range(parts[t.slot]):
                assert (t.slot, 16 * b, part) in seen
    assert len(seen) == sum(t.n_blocks * parts[t.slot] for t in trials)
    for (slot, row0, part), i in seen.items():                    # the kernel finds partner j at i + 8 (j - part)
        for j in range(parts[slot]):
            assert seen[(slot, row0, j)] == i + rp.N_XCD * (j - part)
    if len(seen) <= rp.N_XCD * rp.CUS_PER_XCD - 3 * rp.N_XCD:     # a list that fits the chip leaves nobody in a second round
        assert len(wl) <= rp.N_XCD * rp.CUS_PER_XCD
    # the encoding keeps the three fields apart
    arr = rp.encode(wl, np.dtype([("slot", "<i4"), ("row0", "<i4")]))
    for (slot, row0, part, n), rec in zip(wl, arr):
        assert int(rec["slot"]) == slot and int(rec["row0"]) & 0xFFFFF == row0
        assert (int(rec["row0"]) >> 20) & 15 == part and ((int(rec["row0"]) >> 24) & 15) == (n if n > 1 else 0)
