"""Fused regression-head MLP: ``Linear -> ReLU -> ... -> Linear`` on the last token of each sequence.

The Transformer regressor ends in ``x[:, -1, :]`` followed by ``d -> 128 -> 64 -> 32 -> 16 -> 1`` with ReLUs
(/root/reference/ray-tune-hpo-regression.py:217-221,235-240).  On a GPU that head is one forward and two
backward launches of ``csrc/head_mlp.hip`` -- weights staged through LDS, activations never leaving it, and the
last-token slice folded into the kernel's row addressing; elsewhere ``head_mlp_reference`` runs.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch
import torch.nn.functional as F


def head_mlp_reference(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    h = x
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = F.linear(h, w, b)
        if i < len(weights) - 1:
            h = F.relu(h)
    return h


def _eligible(x: torch.Tensor, weights: Sequence[torch.Tensor]) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        return False
    if not 1 <= len(weights) <= 5 or x.shape[1] > 2048:
        return False
    return all(w.shape[0] <= 128 and w.dtype == torch.float32 for w in weights)


def head_mlp(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """x: [B, d] (any row stride, e.g. the view ``h[:, -1, :]``); returns [B, out]."""
    if _eligible(x, weights):
        return FusedHeadMLP.apply(x, len(weights), *weights, *biases)
    return head_mlp_reference(x, weights, biases)


class FusedHeadMLP(torch.autograd.Function):
    @staticmethod
    def _desc(x_dim, ws, bs, dws=None, dbs=None):
        from . import lib as L
        hd = L.HeadMlp()
        n = len(ws)
        hd.n_layers = n
        hd.dims[0] = x_dim
        for i, w in enumerate(ws):
            hd.dims[i + 1] = w.shape[0]
            hd.W[i], hd.b[i] = w.data_ptr(), bs[i].data_ptr()
            if dws is not None:
                hd.dW[i], hd.db[i] = dws[i].data_ptr(), dbs[i].data_ptr()
        hd.act_stride = sum(w.shape[0] for w in ws[:-1])
        hd.dz_stride = sum(w.shape[0] for w in ws)
        return hd

    @staticmethod
    def forward(ctx, x, n, *params):
        from . import lib as L
        so = L.load()
        ws = [p.contiguous() for p in params[:n]]
        bs = [p.contiguous() for p in params[n:]]
        for i, w in enumerate(ws):
            if w.shape[1] != (x.shape[1] if i == 0 else ws[i - 1].shape[0]) or bs[i].shape[0] != w.shape[0]:
                raise ValueError("head MLP layer shapes do not chain")
        B = x.shape[0]
        hd = FusedHeadMLP._desc(x.shape[1], ws, bs)
        acts = torch.empty(B, max(hd.act_stride, 1), device=x.device, dtype=torch.float32)
        out = torch.empty(B, ws[-1].shape[0], device=x.device, dtype=torch.float32)
        L.check(so.mt_head_mlp_fwd_launch(C.byref(hd), x.data_ptr(), x.stride(0), acts.data_ptr(), out.data_ptr(), B,
                                          L.raw_stream(x.device)), "mt_head_mlp_fwd_launch")
        ctx.save_for_backward(x, acts, *ws, *bs)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import lib as L
        so = L.load()
        n = ctx.n
        x, acts = ctx.saved_tensors[:2]
        ws, bs = list(ctx.saved_tensors[2:2 + n]), list(ctx.saved_tensors[2 + n:])
        B = x.shape[0]
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty_like(b) for b in bs]
        hd = FusedHeadMLP._desc(x.shape[1], ws, bs, dws, dbs)
        dz = torch.empty(B, hd.dz_stride, device=x.device, dtype=torch.float32)
        dx = torch.empty(B, x.shape[1], device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        L.check(so.mt_head_mlp_bwd_launch(C.byref(hd), x.data_ptr(), x.stride(0), acts.data_ptr(),
                                          dout.contiguous().data_ptr(), dz.data_ptr(),
                                          dx.data_ptr() if dx is not None else None, x.shape[1], B,
                                          L.raw_stream(x.device)), "mt_head_mlp_bwd_launch")
        return (dx, None, *dws, *dbs)
