#!/usr/bin/env python
"""Target of the data-parallel overlap trace: one DP trial on ONE GPU with the per-bucket all-reduces forced on.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 bench/dp_overlap_target.py

With one rank RCCL turns every out-of-place all-reduce into a device copy on the communication stream, so the trace shows
what the step driver's event / stream choreography does with real bytes: bucket l's copy starts when its weight gradient
is complete and runs while the input gradient of layer l (and the backward of the layers below) run on the compute
stream.  ``bench/dp_overlap_summary.py`` reduces the trace to a table.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=8192)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    from mitune.options import OPTIONS
    from mitune.parallel.dp import DPTrainer
    OPTIONS.dp_force_collectives = True
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    n = args.batch * (args.steps + 1)
    X = torch.randn(n, 256, device=dev, generator=g)
    y = torch.tanh(X[:, 0]) + 0.1 * X[:, 1]
    cfg = {"hidden_sizes": [args.width] * 3, "batch_size": args.batch, "learning_rate": 1e-4, "optimizer": "adamw",
           "dtype": args.dtype}
    tr = DPTrainer(cfg, X[args.batch:], y[args.batch:], X[:args.batch], y[:args.batch], seed=0)
    tr.steps_per_epoch = args.steps
    for _ in range(2):                      # first pass warms the kernels up, the second is the one to read
        loss = tr.train_epoch()
        torch.cuda.synchronize()
    print(f"loss {loss:.5f}, {tr.reducer.bytes_reduced / 1e9:.2f} GB through the collectives", flush=True)
    tr.close()


if __name__ == "__main__":
    main()
