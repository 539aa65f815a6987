"""Bindings of ``csrc/batch.hip``: batch assembly at a device-resident cursor and the fused mean-squared-error loss of
the single-trial captured step (/root/reference/ray-tune-hpo-regression-sample.py:110-120 does these with a DataLoader
batch, ``nn.MSELoss`` and a ``loss.item()`` per step)."""
from __future__ import annotations

import torch

from . import lib as L


def cursor_batch(perm: torch.Tensor, pos: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, B: int):
    """``X[perm[pos:pos+B]]``, ``Y[perm[pos:pos+B]]`` in one launch; ``perm`` and ``pos`` are int64 tensors on the device."""
    so = L.load()
    xb = torch.empty((B,) + tuple(X.shape[1:]), device=X.device, dtype=torch.float32)
    yb = torch.empty((B,) + tuple(Y.shape[1:]), device=Y.device, dtype=torch.float32)
    xe = X[0].numel()
    ye = max(1, Y[0].numel())
    L.check(so.mt_cursor_batch_launch(perm.data_ptr(), pos.data_ptr(), perm.numel(), X.data_ptr(), Y.data_ptr(), X.shape[0],
                                      xe, ye, xb.data_ptr(), yb.data_ptr(), B,
                                      L.raw_stream(X.device)), "mt_cursor_batch_launch")
    return xb, yb


def cursor_batch_ok(perm: torch.Tensor, pos: torch.Tensor, X: torch.Tensor, Y: torch.Tensor) -> bool:
    return (X.is_cuda and X.dtype == torch.float32 and Y.dtype == torch.float32 and X.is_contiguous() and Y.is_contiguous()
            and perm.dtype == torch.int64 and pos.dtype == torch.int64 and perm.is_contiguous() and X.shape[0] == Y.shape[0])


class FusedMSE(torch.autograd.Function):
    """``mean((out - y)^2)``; the same launch stores the gradient, adds the loss to ``running`` (a device scalar) and, if
    given, advances ``cursor`` (int64 [1]) by ``cursor_inc``."""

    @staticmethod
    def forward(ctx, out, y, running, cursor=None, cursor_inc=0):
        so = L.load()
        oc, yc = out.contiguous(), y.contiguous()
        grad = torch.empty_like(oc)
        loss = torch.empty((), device=oc.device, dtype=torch.float32)
        L.check(so.mt_mse_fwd_launch(oc.data_ptr(), yc.data_ptr(), oc.numel(), grad.data_ptr(), loss.data_ptr(),
                                     running.data_ptr() if running is not None else None,
                                     cursor.data_ptr() if cursor is not None else None, int(cursor_inc),
                                     L.raw_stream(oc.device)), "mt_mse_fwd_launch")
        ctx.save_for_backward(grad)
        ctx.shape = out.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None, None, None
