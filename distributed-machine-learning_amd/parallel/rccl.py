"""A direct RCCL communicator for collectives enqueued from native code on a HIP stream of our choosing.

``torch.distributed`` owns its communicators and its streams: a collective issued through it runs on the process
group's internal stream and comes back to Python as a work handle.  The data-parallel step driver
(``ops/csrc/dp.hip``) enqueues a whole optimiser step from C++ -- GEMMs, bucket casts, all-reduces, optimiser -- and
therefore needs ``ncclAllReduce`` as a plain function it can call with a stream argument.  This module binds the RCCL
library that is ALREADY loaded into the process (the one PyTorch-ROCm ships and uses for its ``nccl`` backend: loading
a second copy would give two runtimes fighting over the same links), creates one communicator per process
(``ncclGetUniqueId`` on rank 0, the 128 id bytes shipped to the others over the existing ``Comm``) and exposes

* ``fn_all_reduce`` / ``comm_handle``: the raw function address and communicator pointer handed to the C++ driver,
* ``all_reduce`` / ``reduce_scatter`` / ``all_gather``: the same calls from Python on a given stream (tests, and the
  measurement of which form is faster for a bucket: on fully connected xGMI, reduce-scatter + all-gather drives all
  seven links of a GPU at once where a single ring is bound by one link).

Nothing here needs a second GPU to be importable or constructible: a one-rank communicator works on one device, which
is what the GPU tests exercise; multi-rank runs happen on the node.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

NCCL_UNIQUE_ID_BYTES = 128
# enum values of rccl.h (ncclDataType_t / ncclRedOp_t)
DTYPES = {torch.int8: 0, torch.uint8: 1, torch.int32: 2, torch.int64: 4, torch.float16: 6, torch.float32: 7,
          torch.float64: 8, torch.bfloat16: 9}
OPS = {"sum": 0, "prod": 1, "max": 2, "min": 3, "avg": 4}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * NCCL_UNIQUE_ID_BYTES)]     # (a c_char array would read back cut at its first zero byte)


class RcclError(RuntimeError):
    pass


def library_path() -> str:
    """The RCCL shared object PyTorch itself links against (falls back to the ROCm installation's)."""
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"]
    for p in cand:
        if os.path.exists(p):
            return p
    raise RcclError(f"librccl.so not found (looked at {cand})")


def _load() -> C.CDLL:
    lib = C.CDLL(library_path())
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclReduceScatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetErrorString.restype = C.c_char_p
    for fn in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclCommDestroy, lib.ncclAllReduce, lib.ncclReduceScatter,
               lib.ncclAllGather):
        fn.restype = C.c_int
    return lib


class RcclComm:
    """One RCCL communicator over the ranks of ``comm`` (any ``parallel.comm.Comm``: used once, to ship the id)."""

    def __init__(self, comm, device: Optional[torch.device] = None):
        self.rank, self.world = int(comm.rank), int(comm.world_size)
        self.device = torch.device(device if device is not None else comm.device)
        if self.device.type != "cuda":
            raise RcclError("RCCL communicators live on GPUs")
        torch.cuda.set_device(self.device)
        from .comm import single_node_bootstrap
        single_node_bootstrap(self.world)
        self.lib = _load()
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = C.string_at(C.byref(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else None
        if self.world > 1:
            raw = comm.broadcast_object(raw, src=0)
        C.memmove(C.byref(uid), raw, NCCL_UNIQUE_ID_BYTES)
        handle = C.c_void_p()
        self._check(self.lib.ncclCommInitRank(C.byref(handle), self.world, uid, self.rank), "ncclCommInitRank")
        self.comm_handle = handle.value
        self.fn_all_reduce = C.cast(self.lib.ncclAllReduce, C.c_void_p).value
        self.stream = torch.cuda.Stream(self.device)        # the communication stream of the step driver

    def _check(self, code: int, what: str) -> None:
        if code != 0:
            raise RcclError(f"{what} failed: {self.lib.ncclGetErrorString(code).decode()} ({code})")

    def _stream(self, stream) -> int:
        return (stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream

    def all_reduce(self, t: torch.Tensor, op: str = "sum", stream=None) -> torch.Tensor:
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), DTYPES[t.dtype], OPS[op], self.comm_handle,
                                           self._stream(stream)), "ncclAllReduce")
        return t

    def reduce_scatter(self, src: torch.Tensor, dst: torch.Tensor, op: str = "sum", stream=None) -> torch.Tensor:
        """``dst`` (numel = src.numel / world) receives this rank's reduced slice of ``src``."""
        if src.numel() != dst.numel() * self.world:
            raise ValueError("reduce_scatter needs src.numel() == world * dst.numel()")
        self._check(self.lib.ncclReduceScatter(src.data_ptr(), dst.data_ptr(), dst.numel(), DTYPES[src.dtype], OPS[op],
                                               self.comm_handle, self._stream(stream)), "ncclReduceScatter")
        return dst

    def all_gather(self, src: torch.Tensor, dst: torch.Tensor, stream=None) -> torch.Tensor:
        if dst.numel() != src.numel() * self.world:
            raise ValueError("all_gather needs dst.numel() == world * src.numel()")
        self._check(self.lib.ncclAllGather(src.data_ptr(), dst.data_ptr(), src.numel(), DTYPES[src.dtype], self.comm_handle,
                                           self._stream(stream)), "ncclAllGather")
        return dst

    def all_reduce_rs_ag(self, t: torch.Tensor, op: str = "sum", stream=None) -> torch.Tensor:
        """All-reduce as reduce-scatter + all-gather in place (``t.numel()`` divisible by the world size)."""
        n = t.numel() // self.world
        if n * self.world != t.numel():
            raise ValueError("tensor length must be divisible by the world size")
        mine = t.view(-1)[self.rank * n:(self.rank + 1) * n]
        self.reduce_scatter(t.view(-1), mine, op, stream)
        self.all_gather(mine, t.view(-1), stream)
        return t

    def close(self) -> None:
        if getattr(self, "comm_handle", None):
            torch.cuda.synchronize(self.device)
            self.lib.ncclCommDestroy(self.comm_handle)
            self.comm_handle = None
