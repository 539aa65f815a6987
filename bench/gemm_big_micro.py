"""One large bf16 GEMM (a layer of a data-parallel trial): the eight-wave 256-tile kernel against the 128-tile kernel and the
BLAS library, forward / input gradient / weight gradient, with a numerics check against an f32 product of a sub-block.

    python bench/gemm_big_micro.py [--shapes 8192,8192,8192 16384,8192,8192] [--iters 20]
"""
import argparse
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from mitune.ops import lib as L      # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["4096,4096,4096", "8192,8192,8192", "16384,8192,8192"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sched", type=int, nargs="+", default=[-1], help="-1: chosen by mode, 0: two barriers per K-tile, 1: one; +8 / +16: timing-only ablations")
    ap.add_argument("--band", type=int, nargs="+", default=[4], help="row tiles per band of the tile order")
    args = ap.parse_args()
    so = L.load()
    so.mt_gemm_p8_min_tiles.argtypes = [C.c_int]
    so.mt_gemm_p8_min_tiles.restype = C.c_int
    so.mt_gemm_p8_option.argtypes = [C.c_int, C.c_int]
    dev = torch.device("cuda")
    st = L.raw_stream(dev)
    for shape in args.shapes:
        M, N, K = (int(v) for v in shape.split(","))
        g = torch.Generator(device="cuda").manual_seed(1)
        X = torch.randn(M, K, device=dev, generator=g).bfloat16()             # activations [M][K]
        W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()    # weight [N][K]
        dY = torch.randn(M, N, device=dev, generator=g).bfloat16()
        bias = torch.randn(N, device=dev, generator=g)
        Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dX = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        dW = torch.empty(N * K + N, device=dev, dtype=torch.float32)
        flop = 2.0 * M * N * K

        def fwd():
            L.check(so.mt_gemm_single_launch(0, X.data_ptr(), W.data_ptr(), Y.data_ptr(), bias.data_ptr(), M, N, K, K, K, N, 0,
                                             1, st), "fwd")

        def dgrad():
            L.check(so.mt_gemm_single_launch(1, dY.data_ptr(), W.data_ptr(), dX.data_ptr(), X.data_ptr(), M, K, N, N, K, K, K,
                                             1, st), "dgrad")

        def wgrad():
            L.check(so.mt_wgrad_single_launch(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), dW.data_ptr() + N * K * 4, M, N, K, 1,
                                              N * K + N, st), "wgrad")

        refs = {"fwd": lambda: torch.relu(X[:256].float() @ W[:512].float().t() + bias[:512]),
                "dgrad": lambda: (dY[:256].float() @ W.float()[:, :512]) * (X[:256, :512].float() > 0),
                "wgrad": lambda: dY[:, :256].float().t() @ X[:, :512].float()}
        outs = {"fwd": lambda: Y[:256, :512].float(), "dgrad": lambda: dX[:256, :512].float(),
                "wgrad": lambda: dW[:N * K].view(N, K)[:256, :512]}
        lib_fns = {"fwd": lambda: torch.relu_(torch.addmm(bias.bfloat16(), X, W.t())),
                   "dgrad": lambda: dY @ W, "wgrad": lambda: dY.t() @ X}
        for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
            row = [f"{shape:>18} {name:6}"]
            for label, lim, sched, band in [(f"p8 s{sc} b{bd}", 1, sc, bd) for sc in args.sched for bd in args.band] + [("t128", 0, 0, 4)]:
                so.mt_gemm_p8_min_tiles(lim)
                so.mt_gemm_p8_option(0, sched)
                so.mt_gemm_p8_option(1, band)
                fn()
                torch.cuda.synchronize()
                ref, got = refs[name](), outs[name]()
                err = float((got - ref).abs().max() / ref.abs().max())
                us = timed(fn, args.iters)
                row.append(f"{label} {us:7.1f} us {flop / us / 1e6:6.1f} TF/s err {err:.0e}")
            us = timed(lib_fns[name], args.iters)
            row.append(f"library {us:8.1f} us {flop / us / 1e6:7.1f} TF/s")
            print(" | ".join(row), flush=True)
        if name == "wgrad":
            db_ref = dY.float().sum(0)
            so.mt_gemm_p8_min_tiles(1)
            wgrad()
            torch.cuda.synchronize()
            print(f"{'':18} bias-gradient err {float((dW[N * K:] - db_ref).abs().max() / db_ref.abs().max()):.1e}", flush=True)
    so.mt_gemm_p8_min_tiles(256)
    so.mt_gemm_p8_option(0, -1)
    so.mt_gemm_p8_option(1, 4)


if __name__ == "__main__":
    main()
