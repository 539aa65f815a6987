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
