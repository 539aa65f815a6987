"""Device-resident random state for dropout on the GPU.

A ``Counter`` is an int64 on the device plus a table of the next ``TABLE`` seeds derived from it.  A dropout site
takes the next table entry as its seed (a view: no launch); when the table is used up -- and at the start of every
training step, see ``begin_step`` -- one small launch writes a fresh table from the counter and advances the
counter.  The kernels of ``csrc/dropout.hip`` / ``csrc/posenc.hip`` / attention / LayerNorm hash ``(seed, element
index)``.  Because counter and table live on the device and are advanced *by the issued work itself*, the refill is
capturable in a hipGraph: a replayed training step draws new masks every time, and nothing depends on the
framework's generator (whose graph-registered Philox state misbehaved here once three or more captured graphs were
alive).  A refill writes a *new* table tensor, so seeds saved for a backward pass stay valid.

Every thread has a default counter per device.  A trainer that runs next to others (several trials' graphs replayed
on different streams) owns a counter and makes it current with ``scope(counter)`` around every forward it issues --
captured work then holds *that* counter's addresses, concurrent trials never race on a shared seed, and a trial's
masks do not depend on what else shares the GPU.  ``seed_all(value)`` / ``Counter.seed(value)`` pin a counter,
e.g. to a function of (experiment seed, trial, epoch).

A region captured into a graph must start with ``begin_step()`` (``GraphedTrainer`` does): the refill is then part
of the graph, and replays do not reuse the table of the capture.
"""
from __future__ import annotations

import contextlib
import threading
from typing import Dict, Optional

import torch

TABLE = 32                                              # seeds per refill; a training step of the zoo draws 5..30
_STRIDE = 0x9E3779B97F4A7C15                            # odd: successive seeds differ in many bits
_tls = threading.local()


def _i64(v: int) -> int:
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


class Counter:
    __slots__ = ("ctr", "table", "idx", "_steps")

    def __init__(self, device, value: int = 0x1234567):
        self.ctr = torch.full((1,), _i64(value), dtype=torch.int64, device=device)
        self._steps = torch.tensor([_i64((i + 1) * _STRIDE) for i in range(TABLE)], dtype=torch.int64, device=device)
        self.table: Optional[torch.Tensor] = None
        self.idx = TABLE

    def seed(self, value: int) -> None:
        """Pin the counter (stream-ordered); the next draw starts a fresh table."""
        self.ctr.fill_(_i64(value))
        self.idx = TABLE

    def begin_step(self) -> None:
        self.idx = TABLE

    def next(self) -> torch.Tensor:
        """Seed for one dropout site: a [1] int64 view that stays valid for the backward pass."""
        if self.idx >= TABLE:
            if self.ctr.is_cuda:                         # one launch: write the table, advance the counter
                from . import lib as L
                self.table = torch.empty(TABLE, dtype=torch.int64, device=self.ctr.device)
                L.check(L.load().mt_seed_refill_launch(self.ctr.data_ptr(), self.table.data_ptr(), TABLE, _i64(_STRIDE),
                                                       L.raw_stream(self.ctr.device)), "mt_seed_refill_launch")
            else:
                self.table = self.ctr + self._steps      # wraps modulo 2^64
                self.ctr.add_(_i64(TABLE * _STRIDE))
            self.idx = 0
        s = self.table[self.idx:self.idx + 1]
        self.idx += 1
        return s


def new_counter(device, value: int = 0x1234567) -> Counter:
    return Counter(device, value)


def counter(device) -> Counter:
    cur: Optional[Counter] = getattr(_tls, "cur", None)
    if cur is not None:
        return cur
    device = torch.device(device)
    if device.index is None and device.type == "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    table: Optional[Dict[torch.device, Counter]] = getattr(_tls, "default", None)
    if table is None:
        table = _tls.default = {}
    c = table.get(device)
    if c is None:
        main = threading.current_thread() is threading.main_thread()
        c = table[device] = Counter(device, 0x1234567 if main else 0x1234567 + threading.get_ident() * 0x51ED27)
    return c


@contextlib.contextmanager
def scope(ctr: Optional[Counter]):
    """Make ``ctr`` the counter every dropout site of this thread uses (None: leave the current one in place)."""
    if ctr is None:
        yield
        return
    prev = getattr(_tls, "cur", None)
    _tls.cur = ctr
    try:
        yield
    finally:
        _tls.cur = prev


def seed_all(value: int, device="cuda") -> None:
    counter(device).seed(value)


def begin_step(device="cuda") -> None:
    counter(device).begin_step()


def next_seed(device) -> torch.Tensor:
    return counter(device).next()
