"""Every tuning knob of the engines in one place.

``OPTIONS`` is the process-wide ``EngineOptions`` instance.  Its defaults are the measured choices; a run changes them
either in code (``mitune.options.OPTIONS.rowblock = "0"`` before the first engine is built, or ``with
options.override(rowblock="0"): ...`` in tests and ablations) or through the one environment variable
``MITUNE_OPTIONS="rowblock=0;f32_chain_rule=128:4,32:2"`` (``;``-separated ``field=value`` pairs, parsed by the field's
type; an unknown field is an error, so a typo cannot silently leave a default in force).

The only other environment variables the package reads are about the launch, not about kernels:
``MITUNE_COMM_BACKEND`` (bench scripts: ``gloo`` rehearses multi-process runs on fewer GPUs than ranks),
``MITUNE_RENDEZVOUS_DIR`` (where surviving ranks meet after a rank is lost) and ``MITUNE_CPU_THREADS`` (CPU threads per
rank).
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
from typing import Any, Dict, Iterator

ENV = "MITUNE_OPTIONS"


@dataclasses.dataclass
class EngineOptions:
    # ---- packs of MLP trials on grouped launches (packed/engine_hip.py)
    big_min_tiles: int = 256          # bf16 packs: a launch takes 256x256 tiles when it has at least this many (0: never)
    big_cost: float = 3.5             # ... and a 256-tile is priced at this many 128-tiles when padding is compared
    ksplits_small: int = 1            # weight-gradient splits over the batch for packs of at most 32 / 64 tiles (4 / 2 at most); 1 = never:
                                      # the split count follows the PACK's size, so anything else lets a trial's rounding depend on its neighbours
    f32_small_tiles: int = 2048       # exact-f32 packs: launches with fewer 128-tiles than this use 64x64 tiles
    f32_tiny_tiles: int = 1024        # ... and launches with fewer 64-tiles than this use 32x32 tiles
    f32_chain_rule: str = "128:4,32:2"   # "min_slots:chains,...": independent chains of launches on separate HIP streams
    chains_by_batch: bool = True      # one chain per batch size (a chain's step count is that of its smallest batch)
    streams: int = 0                  # force this many chains for every pack (0: the rules above)
    rowblock: str = "auto"            # exact-f32 row-block step: "auto" (chains of <= rowblock_max_blocks blocks), "1", "0"
    rowblock_max_blocks: int = 400
    rowblock_split: int = 4           # most workgroups that may share one 16-row block, each taking 1/n of every layer's columns (1: never split)
    rowblock_split_cost: int = 50000  # what one exchange between the parts of a split block costs, in multiply-adds of a 16-row chain
    rowblock_wg_budget: int = 256     # a chain is only split while its work list stays within this many workgroups (one per CU)
    chain_debug: bool = False         # print every chain's composition when a plan is built
    # ---- packs of sequence-model trials on grouped launches (packed/engine_seq.py)
    seq_engine: bool = True           # False keeps sample-Transformer packs on one captured step per trial
    xf_engine: bool = True            # False keeps packs of the configurable Transformer on one captured step per trial
    xf_tiny_tiles: int = 512          # program engine: launches with fewer 64-tiles than this use 32x32 tiles
    xf_bf16_big_tiles: int = 256      # program engine, bf16 products: launches with at least this many 128-tiles take them
    seq_prep_threads: int = 2         # host threads drawing the next trials' initial weights
    seq_ksplits: int = 2              # weight-gradient slabs over the token rows
    seq_chain_rule: str = "32:2"
    attn_exact: int = 2               # f32 attention: 2 = every product in f32 up to 128 tokens, 1 = up to 32, 0 = bf16 matrix form
    # ---- one captured step per trial (packed/engine_graph.py, train/graphed.py)
    graph_streams: int = 4            # HIP streams the slots of a pack are replayed on
    graph_force_streams: bool = False  # use them for packs of big steps too
    in_flight: int = 8                # graph replays the host may run ahead per stream
    eval_graphs: bool = True          # validation passes as captured graphs
    fast_batch: bool = True           # batches by cursor over a device-side permutation instead of index tensors
    # ---- dense layers of one-trial-at-a-time models (ops/linear.py)
    wgrad_workgroups: int = 512       # workgroups a weight-gradient launch is split towards
    small_linear_tile: int = 0        # 0: the launcher picks 16x16 or 32x32 tiles; 16 / 32 force one
    # ---- data-parallel trainer (parallel/dp.py)
    dp_force_collectives: bool = False   # issue the per-bucket all-reduces even with one rank (profiling the overlap)

    def update(self, **kw: Any) -> None:
        fields = {f.name: f for f in dataclasses.fields(self)}
        for k, v in kw.items():
            if k not in fields:
                raise KeyError(f"unknown engine option {k!r}; known: {', '.join(sorted(fields))}")
            setattr(self, k, _coerce(fields[k].type, v))

    def as_dict(self) -> Dict[str, Any]:
        return dataclasses.asdict(self)


def _coerce(kind: Any, v: Any) -> Any:
    kind = {"int": int, "float": float, "bool": bool, "str": str}.get(kind, kind)
    if kind is bool:
        return v if isinstance(v, bool) else str(v).strip().lower() not in ("0", "false", "no", "off", "")
    return kind(v)


def parse(text: str) -> Dict[str, str]:
    out: Dict[str, str] = {}
    for part in (text or "").split(";"):
        part = part.strip()
        if not part:
            continue
        if "=" not in part:
            raise ValueError(f"{ENV}: expected field=value, got {part!r}")
        k, v = part.split("=", 1)
        out[k.strip()] = v.strip()
    return out


OPTIONS = EngineOptions()
OPTIONS.update(**parse(os.environ.get(ENV, "")))


@contextlib.contextmanager
def override(**kw: Any) -> Iterator[EngineOptions]:
    """Temporarily change fields of ``OPTIONS`` (engines built inside the block see them)."""
    before = {k: getattr(OPTIONS, k) for k in kw}
    OPTIONS.update(**kw)
    try:
        yield OPTIONS
    finally:
        for k, v in before.items():
            setattr(OPTIONS, k, v)
