"""Phase timers for the driver loop (host side) and device-event timers for kernels.

The reference has no instrumentation beyond per-epoch loss prints (/root/reference/ray-tune-hpo-regression.py:350).
``PhaseTimer`` accumulates wall time per named phase with negligible overhead; ``DeviceTimer`` brackets GPU
work with events so that a measurement never forces a host synchronisation inside the measured region.
"""
from __future__ import annotations

import contextlib
import time
from collections import defaultdict
from typing import Dict, Iterator, List, Tuple


class PhaseTimer:
    def __init__(self):
        self.total: Dict[str, float] = defaultdict(float)
        self.count: Dict[str, int] = defaultdict(int)

    @contextlib.contextmanager
    def phase(self, name: str) -> Iterator[None]:
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.total[name] += time.perf_counter() - t0
            self.count[name] += 1

    def summary(self) -> Dict[str, Dict[str, float]]:
        return {k: {"total_s": v, "calls": self.count[k], "mean_ms": 1e3 * v / max(self.count[k], 1)}
                for k, v in sorted(self.total.items(), key=lambda kv: -kv[1])}

    def report(self) -> str:
        rows = [f"{k:<24} {v['total_s']:9.3f} s  {v['calls']:>7} calls  {v['mean_ms']:9.3f} ms/call"
                for k, v in self.summary().items()]
        return "\n".join(rows)


class DeviceTimer:
    """Event pairs on the current stream; ``results()`` synchronises once and returns milliseconds."""

    def __init__(self):
        self._pairs: List[Tuple[str, object, object]] = []

    @contextlib.contextmanager
    def span(self, name: str) -> Iterator[None]:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        try:
            yield
        finally:
            e1.record()
            self._pairs.append((name, e0, e1))

    def results(self) -> Dict[str, float]:
        import torch
        torch.cuda.synchronize()
        out: Dict[str, float] = defaultdict(float)
        for name, e0, e1 in self._pairs:
            out[name] += e0.elapsed_time(e1)
        self._pairs.clear()
        return dict(out)
