#!/usr/bin/env python
"""Backward of small f32 dense layers: fused one-launch kernel vs the library's three launches (dgrad, wgrad, bias sum).
Times are per backward: (forward + backward) minus forward, each measured over hipGraph replays of 20 repetitions
(launch gaps included, host excluded)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitune.ops.linear import SmallLinear        # noqa: E402

SHAPES = [(1600, 10, 32), (1600, 32, 96), (1600, 32, 128), (1600, 128, 32), (1600, 64, 192), (1600, 64, 256),
          (1600, 256, 64), (1600, 128, 384), (1600, 128, 512), (1600, 512, 128), (32, 64, 1), (3200, 64, 256),
          (800, 64, 256), (4800, 256, 768), (1600, 256, 1024),
          (6144, 64, 256), (6144, 128, 512), (6144, 192, 64), (12288, 64, 192), (12288, 128, 128), (24576, 64, 128)]


class _Ctx:
    def save_for_backward(self, *a):
        pass


def bench(fn, reps=20, iters=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        s.synchronize()
    return (time.perf_counter() - t0) / (reps * iters) * 1e6


def main():
    print(f"{'R':>6} {'in':>5} {'out':>5} | library us | fused us | speedup")
    for R, i, o in SHAPES:
        x = torch.randn(R, i, device="cuda", requires_grad=True)
        w = torch.randn(o, i, device="cuda", requires_grad=True)
        b = torch.randn(o, device="cuda", requires_grad=True)
        dy = torch.randn(R, o, device="cuda")

        def fwd():
            with torch.no_grad():
                F.linear(x, w, b)

        def fwd_own():
            with torch.no_grad():
                SmallLinear.forward(_Ctx(), x, w, b)

        def lib():
            torch.autograd.grad(F.linear(x, w, b), (x, w, b), dy)

        def own():
            torch.autograd.grad(SmallLinear.apply(x, w, b), (x, w, b), dy)

        SmallLinear.OWN_FORWARD = True
        f_own = bench(fwd_own)
        SmallLinear.OWN_FORWARD = False          # backward columns: both sides on the library forward
        f, a = bench(fwd), bench(lib)
        SmallLinear.TILE = 32
        c32 = bench(own)
        SmallLinear.TILE = 16
        c16 = bench(own)
        SmallLinear.TILE = 0
        c = bench(own)
        print(f"{R:6d} {i:5d} {o:5d} | {a - f:10.1f} | {c - f:8.1f} | {(a - f) / max(c - f, 1e-3):6.2f}   (forward {f:.1f}; "
              f"32x32 tiles {c32 - f:.1f}, 16x16 tiles {c16 - f:.1f}; own forward {f_own:.1f})", flush=True)


if __name__ == "__main__":
    main()
