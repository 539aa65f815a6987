#!/usr/bin/env python
"""Fixed, short workload of the large single bf16 GEMMs for hardware-counter collection (run under ``rocprofv3 --pmc``):
8192 x 8192 x 8192 forward / input gradient / weight gradient on the eight-wave 256-tile kernel (both K-loop schedules) and
on the 128-tile kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mitune.ops import lib as L       # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    so = L.load()
    so.mt_gemm_p8_option.argtypes = [C.c_int, C.c_int]
    dev = torch.device("cuda")
    st = L.raw_stream(dev)
    M = N = K = 8192
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(M, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    dY = torch.randn(M, N, device=dev, generator=g).bfloat16()
    bias = torch.randn(N, device=dev, generator=g)
    Y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    dX = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    dW = torch.empty(N * K + N, device=dev, dtype=torch.float32)
    for lim, sched in ((1, 0), (1, 1), (0, 0)):
        so.mt_gemm_p8_min_tiles(lim)
        so.mt_gemm_p8_option(0, sched)
        for _ in range(iters):
            so.mt_gemm_single_launch(0, X.data_ptr(), W.data_ptr(), Y.data_ptr(), bias.data_ptr(), M, N, K, K, K, N, 0, 1, st)
            so.mt_gemm_single_launch(1, dY.data_ptr(), W.data_ptr(), dX.data_ptr(), X.data_ptr(), M, K, N, N, K, K, K, 1, st)
            so.mt_wgrad_single_launch(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), dW.data_ptr() + N * K * 4, M, N, K, 1,
                                      N * K + N, st)
        torch.cuda.synchronize()
    so.mt_gemm_p8_option(0, -1)
    so.mt_gemm_p8_min_tiles(256)
    for _ in range(iters):
        torch.relu_(torch.addmm(bias.bfloat16(), X, W.t()))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
