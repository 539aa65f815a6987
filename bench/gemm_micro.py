#!/usr/bin/env python
"""Micro-benchmark of the grouped GEMM kernels on pack-shaped problems (random bf16 data, one process).

    python bench/gemm_micro.py [--slots 16 256] [--width 512] [--batch 512] [--iters 200] [--ablate]
    python bench/gemm_micro.py --variants bench/variants/libgemm_*.so        # A/B of several builds

For each pack size it times forward (NT), dgrad (NN) and wgrad (TN) launches over `slots` groups of
[batch x width] x [width x width] and prints microseconds per launch and achieved TFLOP/s.  With
--variants, every listed build of gemm.hip is timed on the SAME buffers in interleaved rounds (median
and min over rounds), which is the only comparison that survives device-to-device clock differences.
"""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mitune.ops import lib as L  # noqa: E402
from mitune.packed.engine_hip import balanced_tiles  # noqa: E402


def up(arr):
    return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).cuda()


class Problem:
    def __init__(self, mode, T, B, W, ksplits=1, edge=128, dtype=torch.bfloat16):
        g = torch.Generator(device="cuda").manual_seed(0)
        self.mode, self.T, self.B, self.W, self.ksplits = mode, T, B, W, ksplits
        x = (torch.randn(T, B, W, device="cuda", generator=g)).to(dtype)
        w = (torch.randn(T, W, W, device="cuda", generator=g) * 0.05).to(dtype)
        dy = (torch.randn(T, B, W, device="cuda", generator=g)).to(dtype)
        y = torch.empty(T, B, W, device="cuda", dtype=dtype)
        bias = torch.zeros(T, W, device="cuda")
        # weight and bias gradients share one arena per split, like the engine's gradient slabs do
        self.slab = T * W * W + T * W
        arena = torch.empty(ksplits, self.slab, device="cuda")
        dw = arena[:, :T * W * W].view(ksplits, T, W, W)
        db = arena[:, T * W * W:].view(ksplits, T, W)
        groups = np.zeros(T, dtype=L.GEMM_GROUP_DTYPE)
        tl = []
        for t in range(T):
            if mode == 0:
                groups[t] = L.gemm_group(x[t].data_ptr(), w[t].data_ptr(), y[t].data_ptr(), bias[t].data_ptr(), 0, B, W, W, W, W, W, 0, 1)
                tl.append((t, -(-B // edge), -(-W // edge), W // 64))
            elif mode == 1:
                groups[t] = L.gemm_group(dy[t].data_ptr(), w[t].data_ptr(), y[t].data_ptr(), x[t].data_ptr(), 0, B, W, W, W, W, W, W, 1)
                tl.append((t, -(-B // edge), -(-W // edge), W // 64))
            else:
                groups[t] = L.gemm_group(dy[t].data_ptr(), x[t].data_ptr(), dw[0, t].data_ptr(), 0, db[0, t].data_ptr(), W, W, B, W, W, W, 0, 1)
                tl.append((t, -(-W // edge), -(-W // edge), B // 64))
        tiles = balanced_tiles(tl, ksplits if mode == 2 else 1)
        self.keep = (x, w, dy, y, bias, arena)
        self.gt, self.tt, self.ntiles = up(groups), up(tiles), len(tiles)
        self.flop = 2.0 * T * B * W * W

    def time(self, launch_fn, iters, flags=1, nstage=0):
        stream = torch.cuda.current_stream().cuda_stream
        ks = self.ksplits if self.mode == 2 else 1
        stride = self.slab if self.mode == 2 else 0

        def launch():
            rc = launch_fn(self.mode, self.gt.data_ptr(), self.tt.data_ptr(), self.ntiles, 0, flags, 0, 0, ks, stride,
                           nstage, stream)
            assert rc == 0, rc

        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters


def typed(fn):
    fn.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                   C.c_int, C.c_void_p]
    fn.restype = C.c_int
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, nargs="+", default=[16, 64, 256])
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--ablate", action="store_true", help="also time variants with MFMA / loads / epilogue removed")
    ap.add_argument("--variants", nargs="*", default=None, help="shared objects built from gemm.hip to A/B")
    ap.add_argument("--f32", action="store_true", help="time the exact-f32 kernels (gemm_f32.hip) with 128- and 64-tiles")
    args = ap.parse_args()
    names = {0: "fwd  NT", 1: "dgrad NN", 2: "wgrad TN"}
    if args.variants:
        libs = [(os.path.basename(p).replace("libgemm_", "").replace(".so", ""),
                 typed(C.CDLL(os.path.abspath(p)).mt_gemm_launch)) for p in args.variants]
        for T in args.slots:
            for mode in (0, 1, 2):
                prob = Problem(mode, T, args.batch, args.width)
                res = {n: [] for n, _ in libs}
                for _ in range(args.rounds):
                    for n, fn in libs:
                        res[n].append(prob.time(fn, args.iters))
                line = "  ".join(f"{n}: med {statistics.median(v):7.2f} min {min(v):7.2f}" for n, v in res.items())
                print(f"slots {T:4d} {names[mode]} | {line}", flush=True)
        return
    so = L.load()
    if args.f32:
        fn = so.mt_gemm_f32_launch
        for T in args.slots:
            for mode in (0, 1, 2):
                for ks in ((1, 4) if mode == 2 else (1,)):
                    p128 = Problem(mode, T, args.batch, args.width, ks, edge=128, dtype=torch.float32)
                    p64 = Problem(mode, T, args.batch, args.width, ks, edge=64, dtype=torch.float32)
                    t128 = min(p128.time(fn, args.iters, 1, 128) for _ in range(3))
                    t64 = min(p64.time(fn, args.iters, 1, 64) for _ in range(3))
                    t66 = min(p64.time(fn, args.iters, 1, 65) for _ in range(3))       # 64-tiles, three-stage pipeline
                    p32 = Problem(mode, T, args.batch, args.width, ks, edge=32, dtype=torch.float32)
                    t32 = min(p32.time(fn, args.iters, 1, 32) for _ in range(3))
                    t64d = min(p64.time(fn, args.iters, 1, 66) for _ in range(3))      # 64-tiles, four stages, one barrier
                    t32d = min(p32.time(fn, args.iters, 1, 33) for _ in range(3))      # 32-tiles, eight stages, one barrier
                    ku = {c: min((p64 if c == 67 else p32).time(fn, args.iters, 1, c) for _ in range(3)) for c in (35,)}
                    print(f"f32 slots {T:4d}  {names[mode]}  ksplits {ks}  batch {args.batch} width {args.width}  "
                          f"128-tile {t128:8.2f} us ({p128.flop / t128 / 1e6:6.1f} TF/s, {p128.ntiles} tiles)  "
                          f"64-tile {t64:8.2f} us ({p64.flop / t64 / 1e6:6.1f} TF/s, {p64.ntiles} tiles)  "
                          f"64-tile 3-stage {t66:8.2f} us ({p64.flop / t66 / 1e6:6.1f} TF/s)  "
                          f"32-tile {t32:8.2f} us ({p32.flop / t32 / 1e6:6.1f} TF/s, {p32.ntiles} tiles)  "
                          f"64-tile 4-stage {t64d:8.2f} us  32-tile 8-stage {t32d:8.2f} us  "
                          f"32-tile 8-stage 4 K-steps per barrier {ku[35]:7.2f} us", flush=True)
        return
    fn = so.mt_gemm_launch
    for T in args.slots:
        for mode in (0, 1, 2):
            for ks in ((1, 4) if mode == 2 else (1,)):
                prob = Problem(mode, T, args.batch, args.width, ks)
                big = Problem(mode, T, args.batch, args.width, ks, edge=256)
                t2 = [prob.time(fn, args.iters, 1, 2) for _ in range(3)]
                t4 = [prob.time(fn, args.iters, 1, 4) for _ in range(3)]
                tb = [big.time(fn, args.iters, 1, 256) for _ in range(3)]
                t8 = [big.time(fn, args.iters, 1, 258) for _ in range(3)]      # 256-tiles on eight waves, pipelined reads
                us = min(min(t2), min(t4), min(tb), min(t8))
                print(f"slots {T:4d}  {names[mode]}  ksplits {ks}  tiles {prob.ntiles:5d}  2-stage {min(t2):8.2f} us  "
                      f"4-stage {min(t4):8.2f} us  256-tile {min(tb):8.2f} us  256-tile 8-wave {min(t8):8.2f} us  best {prob.flop / us / 1e6:8.1f} TFLOP/s",
                      flush=True)
                if args.ablate and ks == 1:
                    for label, fl in (("no-mfma", 1 | 256), ("no-steady-loads", 1 | 512), ("no-epilogue", 1 | 1024),
                                      ("loads-only", 1 | 256 | 1024), ("mfma-only", 1 | 512 | 1024)):
                        print(f"        ablation {label:<16} {prob.time(fn, args.iters, fl):9.2f} us", flush=True)


if __name__ == "__main__":
    main()
