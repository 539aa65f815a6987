"""Point-to-point transfer of a trial's state between ranks (PBT exploit across GPUs).

A trial's exported state is ``{"model": state_dict, "optimizer": ..., "epoch", "step"}``.  Donor and
receiver agree on the tensor layout because the receiver passes its own exported state as the
template (an exploit copies between trials of the same architecture family; when the shapes differ
the donor's architecture wins and the template is rebuilt from a broadcast header).  All tensors are
flattened into one buffer so the transfer is a single ``send``/``recv`` -- one large xGMI message
rather than one small message per parameter.
"""
from __future__ import annotations

import io
from typing import Any, Dict, List, Tuple

import torch

from .comm import Comm


def _tensor_items(state: Dict, prefix: str = "") -> List[Tuple[str, torch.Tensor]]:
    out: List[Tuple[str, torch.Tensor]] = []
    for k in sorted(state, key=str):
        v = state[k]
        name = f"{prefix}{k}"
        if isinstance(v, torch.Tensor):
            out.append((name, v))
        elif isinstance(v, dict):
            out.extend(_tensor_items(v, name + "\x1f"))
    return out


def _strip_tensors(state: Any) -> Any:
    if isinstance(state, torch.Tensor):
        return {"__tensor__": list(state.shape), "dtype": str(state.dtype).replace("torch.", "")}
    if isinstance(state, dict):
        return {k: _strip_tensors(v) for k, v in state.items()}
    if isinstance(state, (list, tuple)):
        return [_strip_tensors(v) for v in state]
    return state


def _fill_tensors(skel: Any, flat: torch.Tensor, cursor: List[int]) -> Any:
    if isinstance(skel, dict) and "__tensor__" in skel:
        shape = skel["__tensor__"]
        n = 1
        for s in shape:
            n *= s
        t = flat[cursor[0]:cursor[0] + n].reshape(shape).to(getattr(torch, skel["dtype"])).clone()
        cursor[0] += n
        return t
    if isinstance(skel, dict):
        return {k: _fill_tensors(skel[k], flat, cursor) for k in skel}
    if isinstance(skel, list):
        return [_fill_tensors(v, flat, cursor) for v in skel]
    return skel


def _flatten_in_order(state: Any, out: List[torch.Tensor]) -> None:
    if isinstance(state, torch.Tensor):
        out.append(state.detach().reshape(-1).to(torch.float64))
    elif isinstance(state, dict):
        for k in state:
            _flatten_in_order(state[k], out)
    elif isinstance(state, (list, tuple)):
        for v in state:
            _flatten_in_order(v, out)


def send_state(comm: Comm, state: Dict, dst: int) -> None:
    import pickle
    header = pickle.dumps(_strip_tensors(state))
    hbuf = torch.frombuffer(bytearray(header), dtype=torch.uint8).to(torch.float64)
    parts: List[torch.Tensor] = []
    _flatten_in_order(state, parts)
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float64)
    comm.send(torch.tensor([hbuf.numel(), flat.numel()], dtype=torch.float64), dst)
    comm.send(hbuf, dst)
    comm.send(flat, dst)


def recv_state(comm: Comm, template: Dict, src: int) -> Dict:
    import pickle
    sizes = comm.recv(torch.zeros(2, dtype=torch.float64), src)
    hbuf = comm.recv(torch.zeros(int(sizes[0]), dtype=torch.float64), src)
    flat = comm.recv(torch.zeros(int(sizes[1]), dtype=torch.float64), src)
    skel = pickle.loads(bytes(hbuf.to(torch.uint8).tolist()))
    return _fill_tensors(skel, flat, [0])
