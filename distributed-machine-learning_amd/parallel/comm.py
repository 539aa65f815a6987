"""Collective-communication interface with RCCL, gloo and in-process backends.

The reference has no communication layer of its own -- trial results travel through Ray's control
plane as Python objects (/root/reference/ray-tune-hpo-regression.py:373,470).  Here every rank
(one process per GPU) runs the *same* scheduler on the *same* all-gathered metric table, so the only
traffic is tensors:

* ``all_gather``  - per-epoch metric records of every rank's trials (a few KB, latency-bound);
* ``all_reduce``  - gradients of data-parallel oversized trials (bandwidth-bound, bucketed);
* ``send``/``recv``/``broadcast`` - parameter slabs for PBT exploits across GPUs.

``TorchDistComm`` sits on ``torch.distributed``: backend ``"nccl"`` is RCCL on ROCm and runs over
xGMI between the 8 GPUs of a node; ``"gloo"`` serves CPU tests.  ``ThreadComm`` wires N ranks of one
process together through shared lists (no sockets) for fast unit tests of replicated logic.
"""
from __future__ import annotations

import json
import os
import tempfile
import threading
import time
from typing import Any, List, Optional

import torch


class GatherHandle:
    """A posted all-gather: ``wait()`` completes it (once) and returns the [world, ...] result."""

    def __init__(self, finish):
        self._finish, self._value = finish, None

    def wait(self) -> torch.Tensor:
        if self._finish is not None:
            finish, self._finish = self._finish, None
            self._value = finish()
        return self._value


class Comm:
    rank: int = 0
    world_size: int = 1
    device: torch.device = torch.device("cpu")
    # True when a collective whose peer has died raises in the caller (so that the caller can re-form the group) instead
    # of hanging or ending the process from a watchdog thread
    delivers_errors: bool = True

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """Stack equal-shaped ``t`` from every rank along a new leading dim."""
        raise NotImplementedError

    def all_gather_async(self, t: torch.Tensor) -> "GatherHandle":
        """Post the all-gather of ``t`` and return at once; ``handle.wait()`` gives the stacked result.  Every rank posts
        and waits in the same order.  The default takes a snapshot of ``t`` and meets the other ranks only at
        ``wait()`` (no background progress, same result); backends with asynchronous collectives overlap the transfer
        -- and the wait for slower ranks -- with whatever the caller does in between."""
        snap = t.detach().clone()
        return GatherHandle(lambda: self.all_gather(snap))

    def all_reduce(self, t: torch.Tensor, op: str = "sum", async_op: bool = False):
        raise NotImplementedError

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        raise NotImplementedError

    def send(self, t: torch.Tensor, dst: int) -> None:
        raise NotImplementedError

    def recv(self, t: torch.Tensor, src: int) -> torch.Tensor:
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def broadcast_object(self, obj: Any, src: int = 0) -> Any:
        raise NotImplementedError

    def shrink(self) -> List[int]:
        """After a failed collective: re-form the group from the ranks that are still there.  Every survivor calls it;
        on return ``rank`` / ``world_size`` are the new ones and the result lists the survivors' OLD ranks in new-rank
        order.  Backends that cannot do this raise ``NotImplementedError`` (the job then has to restart)."""
        raise NotImplementedError

    def close(self) -> None:
        pass


class LocalComm(Comm):
    """world_size == 1: every collective is the identity."""

    def __init__(self, device: Optional[torch.device] = None):
        self.rank, self.world_size = 0, 1
        self.device = device or torch.device("cpu")

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_reduce(self, t, op="sum", async_op=False):
        return None

    def broadcast(self, t, src=0):
        return t

    def send(self, t, dst):
        raise RuntimeError("send() with world_size 1")

    def recv(self, t, src):
        raise RuntimeError("recv() with world_size 1")

    def barrier(self):
        pass

    def broadcast_object(self, obj, src=0):
        return obj


def single_node_bootstrap(world_size: int) -> None:
    """RCCL's bootstrap (the TCP side channel that sets a communicator up) picks a network interface by itself; in a
    container whose host name does not resolve, or whose outward interface is filtered, that handshake times out.  When
    every rank of the job is on this node (the launcher's LOCAL_WORLD_SIZE equals the world size, or there is one rank)
    the loopback interface is the right one; an explicit NCCL_SOCKET_IFNAME always wins.
    A launch that exports only RANK / WORLD_SIZE (srun, mpirun, by hand) says nothing about where its ranks are: the
    choice is then left to RCCL."""
    local = os.environ.get("LOCAL_WORLD_SIZE")
    if world_size <= 1 or (local is not None and local != "" and int(local) == world_size):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def file_rendezvous(gdir: str, old_rank: int, old_world: int, settle_s: float, timeout_s: float) -> List[int]:
    """Agree, through files in ``gdir`` alone, on which of ``old_world`` ranks are still there; returns their old ranks in
    ascending order (the same list on every member) or raises on a rank that is not among them.

    Every caller drops ``alive.<rank>``.  When all old ranks have shown up, or nobody new has for ``settle_s`` seconds, the
    lowest rank present publishes the list it sees (``members.json``, created by an atomic link: the first writer wins,
    and whatever it wrote stands for everybody).  A rank that arrives after that finds itself left out and raises: to
    the members it is as lost as the ranks that never came, and they have already taken its work."""
    os.makedirs(gdir, exist_ok=True)
    members_path = os.path.join(gdir, "members.json")
    with open(os.path.join(gdir, f"alive.{old_rank}"), "w") as f:
        f.write(str(os.getpid()))

    def alive() -> List[int]:
        return sorted(int(n.split(".", 1)[1]) for n in os.listdir(gdir) if n.startswith("alive."))

    t0 = time.time()
    seen, last_change = alive(), time.time()
    while not os.path.exists(members_path):
        now_alive = alive()
        if now_alive != seen:
            seen, last_change = now_alive, time.time()
        settled = len(seen) == old_world or time.time() - last_change >= settle_s
        if settled and seen and seen[0] == old_rank:
            tmp = os.path.join(gdir, f"members.tmp.{old_rank}")
            with open(tmp, "w") as f:
                json.dump(seen, f)
            try:
                os.link(tmp, members_path)        # fails if someone else published first: theirs stands
            except FileExistsError:
                pass
            break
        if time.time() - t0 > timeout_s:
            raise RuntimeError(f"rank {old_rank}: no member list appeared in {gdir} within {timeout_s} s")
        time.sleep(0.02)
    while not os.path.exists(members_path):
        time.sleep(0.01)
    with open(members_path) as f:
        members = [int(r) for r in json.load(f)]
    if old_rank not in members:
        raise RuntimeError(f"rank {old_rank} reached the rendezvous after the survivors {members} had re-formed the group")
    return members


class TorchDistComm(Comm):
    """``torch.distributed`` process group (RCCL when the backend is "nccl")."""

    _OPS = {"sum": "SUM", "max": "MAX", "min": "MIN", "avg": "AVG"}

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None,
                 init_method: Optional[str] = None, rank: Optional[int] = None,
                 world_size: Optional[int] = None, timeout_s: float = 600.0, rendezvous_dir: Optional[str] = None,
                 settle_s: float = 3.0):
        import datetime
        import torch.distributed as dist
        self.dist = dist
        self._own = False
        self.timeout_s, self.settle_s, self._generation = float(timeout_s), float(settle_s), 0
        # where the survivors of a lost rank find each other (shrink): a directory every rank of the node can write
        self.rendezvous_dir = rendezvous_dir or os.environ.get("MITUNE_RENDEZVOUS_DIR") or os.path.join(
            tempfile.gettempdir(), f"mitune_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getuid()}")
        if not dist.is_initialized():
            rank = int(os.environ.get("RANK", 0)) if rank is None else rank
            world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl" and device is None:
                local = int(os.environ.get("LOCAL_RANK", rank % max(torch.cuda.device_count(), 1)))
                torch.cuda.set_device(local)
                device = torch.device("cuda", local)
            kw = {}
            if backend == "nccl":
                kw["device_id"] = device
                single_node_bootstrap(world_size)
                # RCCL collectives are asynchronous: left alone, a dead peer shows up in torch's watchdog thread, which
                # ends the process after the group's timeout.  With blocking waits the wait inside the collective call
                # polls the work, raises in THIS thread when the timeout passes or the communicator reports an error,
                # and the watchdog leaves the process alone -- which is what lets PackedExecutor re-form the group
                # (shrink).  The collectives here run once per scheduling round, so a blocking host wait costs nothing.
                os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
            dist.init_process_group(backend=backend, init_method=init_method or "env://", rank=rank,
                                    world_size=world_size, timeout=datetime.timedelta(seconds=timeout_s), **kw)
            self._own = True
        self.backend = dist.get_backend()
        self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        # parity unpinned on RCCL: every test box has one GPU, so a peer lost from a multi-rank RCCL group has never been
        # observed here; the gloo and in-process backends are covered by tests/test_distributed_cpu.py
        self.delivers_errors = self.backend != "nccl" or os.environ.get("TORCH_NCCL_BLOCKING_WAIT", "0") == "1"
        # One process per GPU: the ranks of a node share its CPUs.  Left alone, every rank keeps a full-size intra-op
        # thread pool whose idle workers spin; together they overrun the node's CPU allowance and each small host-side
        # tensor op then stalls for milliseconds (measured with two ranks: 1.2 ms per in-place multiply of a weight
        # matrix, 44 ms per trial load instead of under 1 ms).  The host work of a rank is small tensors: a few threads.
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
        if local > 1:
            want = int(os.environ.get("MITUNE_CPU_THREADS", "0") or 0) or max(1, min(4, torch.get_num_threads() // local))
            torch.set_num_threads(want)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        self.device = device
        if self.world_size > 1:
            # a name for this job's rendezvous that every rank knows before anything fails (a directory left by an
            # earlier job on the same port must not be mistaken for this one's); it travels through the group's
            # key-value store, not through a collective
            import uuid
            try:
                store = dist.distributed_c10d._get_default_store()
                if self.rank == 0:
                    store.set("mitune/rendezvous_token", uuid.uuid4().hex[:12])
                token = store.get("mitune/rendezvous_token").decode()
            except Exception:       # noqa: BLE001 - no store to ask: the port alone names the job
                token = "job"
            self.rendezvous_dir = os.path.join(self.rendezvous_dir, token)

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t if t.device == self.device else t.to(self.device)

    def all_gather(self, t):
        src = self._dev(t.contiguous()).reshape(-1)
        out = torch.empty(self.world_size * src.numel(), dtype=src.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src)       # one collective, ranks concatenated along dim 0
        return out.reshape((self.world_size,) + tuple(t.shape)).to(t.device)

    def all_gather_async(self, t):
        src = self._dev(t.detach().contiguous()).reshape(-1)
        if src.data_ptr() == t.data_ptr():
            src = src.clone()                # the caller may rewrite its tensor while the collective is in flight
        out = torch.empty(self.world_size * src.numel(), dtype=src.dtype, device=src.device)
        work = self.dist.all_gather_into_tensor(out, src, async_op=True)
        shape, back = (self.world_size,) + tuple(t.shape), t.device

        def finish(keep=src):              # holds the send buffer until the collective has completed
            work.wait()
            return out.reshape(shape).to(back)
        return GatherHandle(finish)

    def all_reduce(self, t, op="sum", async_op=False):
        assert t.device == self.device or self.backend != "nccl", "all_reduce tensor must live on the comm device"
        rop = getattr(self.dist.ReduceOp, self._OPS[op])
        if op == "avg" and self.backend != "nccl":
            w = self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, async_op=False)
            t.div_(self.world_size)
            return None
        return self.dist.all_reduce(t, op=rop, async_op=async_op)

    def broadcast(self, t, src=0):
        if t.device != self.device:      # a GPU-resident buffer under the CPU backend: staged through the host
            buf = t.to(self.device)
            self.dist.broadcast(buf, src=src)
            if self.rank != src:
                t.copy_(buf)
            return t
        self.dist.broadcast(t, src=src)
        return t

    def send(self, t, dst):
        self.dist.send(self._dev(t.contiguous()), dst=dst)

    def recv(self, t, src):
        buf = t if t.device == self.device else torch.empty_like(t, device=self.device)
        self.dist.recv(buf, src=src)
        if buf is not t:
            t.copy_(buf)
        return t

    def barrier(self):
        if self.backend == "nccl":
            self.dist.barrier(device_ids=[self.device.index])
        else:
            self.dist.barrier()

    def broadcast_object(self, obj, src=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def close(self):
        if self._own and self.dist.is_initialized():
            self.dist.destroy_process_group()

    def shrink(self) -> List[int]:
        """Survivors of a failed collective meet through files in ``rendezvous_dir`` (the store of the old group lives in
        rank 0's process, which may be the one that died): each drops its old group first -- closing its connections is
        what makes the failure reach ranks that were not talking to the dead one -- then writes ``alive.<old rank>`` into
        a fresh generation directory.  When every old rank has shown up, or nobody new has for ``settle_s`` seconds, the
        lowest rank present publishes the member list (created atomically, first writer wins); a rank that arrives after
        that finds itself left out and raises.  The members then form a new group over a file store in that directory."""
        import datetime
        dist, old_rank, old_world = self.dist, self.rank, self.world_size
        backend = self.backend
        try:
            if dist.is_initialized():
                if backend == "nccl" and hasattr(dist.distributed_c10d, "_abort_process_group"):
                    # an RCCL communicator with a dead peer cannot finish its pending work: abort it instead of
                    # draining it (an orderly destroy would wait for the collective that just failed)
                    dist.distributed_c10d._abort_process_group()
                else:
                    dist.destroy_process_group()
        except Exception:       # noqa: BLE001 - a group with a dead peer may refuse an orderly shutdown
            pass
        self._generation += 1
        gdir = os.path.join(self.rendezvous_dir, f"gen{self._generation}")
        members = file_rendezvous(gdir, old_rank, old_world, self.settle_s, self.timeout_s)
        new_rank, new_world = members.index(old_rank), len(members)
        kw = {}
        if backend == "nccl":
            kw["device_id"] = self.device
            single_node_bootstrap(new_world)
        store = dist.FileStore(os.path.join(gdir, "store"), new_world)
        dist.init_process_group(backend=backend, store=store, rank=new_rank, world_size=new_world,
                                timeout=datetime.timedelta(seconds=self.timeout_s), **kw)
        self._own = True
        self.rank, self.world_size = new_rank, new_world
        return members


class _ThreadHub:
    def __init__(self, n: int):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots: List[Any] = [None] * n
        self.mail = {}
        self.cv = threading.Condition()
        self.dead: set = set()
        self.lock = threading.Lock()
        self.successor: Optional["_ThreadHub"] = None
        self.members: List[int] = []

    def leave(self, rank: int) -> None:
        """A rank vanishes: whoever waits in a collective, now or later, gets an error."""
        with self.lock:
            self.dead.add(rank)
        self.barrier.abort()


class ThreadComm(Comm):
    """N ranks as N threads of one process; collectives go through shared memory + a barrier."""

    def __init__(self, hub: _ThreadHub, rank: int):
        self.hub, self.rank, self.world_size = hub, rank, hub.n
        self.device = torch.device("cpu")

    @staticmethod
    def make(n: int) -> List["ThreadComm"]:
        hub = _ThreadHub(n)
        return [ThreadComm(hub, r) for r in range(n)]

    def _exchange(self, value: Any) -> List[Any]:
        self.hub.slots[self.rank] = value
        self.hub.barrier.wait()
        vals = list(self.hub.slots)
        self.hub.barrier.wait()
        return vals

    def shrink(self) -> List[int]:
        hub = self.hub
        with hub.lock:
            if hub.successor is None:
                hub.members = [r for r in range(hub.n) if r not in hub.dead]
                hub.successor = _ThreadHub(len(hub.members))
            members, new = hub.members, hub.successor
        if self.rank not in members:
            raise RuntimeError(f"rank {self.rank} is not among the survivors {members}")
        self.hub, self.rank, self.world_size = new, members.index(self.rank), len(members)
        return list(members)

    def all_gather(self, t):
        return torch.stack([v.clone() for v in self._exchange(t.detach().clone())])

    def all_reduce(self, t, op="sum", async_op=False):
        vals = torch.stack(self._exchange(t.detach().clone()))
        if op == "sum":
            r = vals.sum(0)
        elif op == "avg":
            r = vals.mean(0)
        elif op == "max":
            r = vals.max(0).values
        else:
            r = vals.min(0).values
        t.copy_(r)
        return None

    def broadcast(self, t, src=0):
        vals = self._exchange(t.detach().clone() if self.rank == src else None)
        t.copy_(vals[src])
        return t

    def send(self, t, dst):
        with self.hub.cv:
            self.hub.mail.setdefault((self.rank, dst), []).append(t.detach().clone())
            self.hub.cv.notify_all()

    def recv(self, t, src):
        with self.hub.cv:
            self.hub.cv.wait_for(lambda: self.hub.mail.get((src, self.rank)))
            t.copy_(self.hub.mail[(src, self.rank)].pop(0))
        return t

    def barrier(self):
        self.hub.barrier.wait()

    def broadcast_object(self, obj, src=0):
        return self._exchange(obj if self.rank == src else None)[src]


def open_comm(prefer: Optional[str] = None, timeout_s: float = 600.0):
    """The rank's communicator and compute device for a job started by a launcher (``WORLD_SIZE`` > 1), or a local
    no-op communicator.  RCCL (backend ``"nccl"``) is tried first; when it cannot form the group -- a node whose driver
    refuses the IPC handles RCCL needs, fewer GPUs than ranks on a rehearsal box -- the job goes on over gloo on the
    next port instead of ending there: what the ranks exchange per round is a table of a few kilobytes, so the
    backend changes what a data-parallel trial costs, not what a packed search measures.  ``prefer`` (or
    ``MITUNE_COMM_BACKEND``) names a backend to use without trying the other.  Every rank takes the same decision:
    group creation with a ``device_id`` is collective and eager, so a failure shows on all of them."""
    import sys
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local % n_dev) if n_dev else torch.device("cpu")
    if world <= 1:
        return LocalComm(device), device
    backend = prefer or os.environ.get("MITUNE_COMM_BACKEND")
    if backend is None:
        # more ranks on this node than GPUs (a rehearsal box): RCCL refuses two ranks on one device, and every rank of the
        # node can tell from the launcher's environment alone, so all of them go to gloo without trying
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
        backend = "nccl" if n_dev and local_world <= n_dev else "gloo"
    if backend == "nccl":
        try:
            comm = TorchDistComm(backend="nccl", timeout_s=timeout_s)
            probe = torch.ones(1, device=comm.device)
            comm.all_reduce(probe)                       # the first collective proves the communicator
            if int(probe.item()) != world:
                raise RuntimeError(f"all-reduce over {world} ranks returned {probe.item()}")
            return comm, comm.device
        except Exception as e:      # noqa: BLE001 - RCCL reports through several exception types
            print(f"[mitune] rank {os.environ.get('RANK', '?')}: RCCL group not formed ({type(e).__name__}: "
                  f"{str(e).splitlines()[0] if str(e) else ''}); continuing over gloo", file=sys.stderr, flush=True)
            import torch.distributed as dist
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:       # noqa: BLE001
                pass
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            backend = "gloo"
    if device.type == "cuda":
        torch.cuda.set_device(device)
    return TorchDistComm(backend=backend, timeout_s=timeout_s), device


def default_comm() -> Comm:
    """RCCL/gloo group when launched under torchrun (WORLD_SIZE > 1), else a local no-op comm."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return TorchDistComm()
    return LocalComm(torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu"))
