#!/usr/bin/env python
"""Late-ASHA regime micro-benchmark: a small pack (default 16 wide trials) stepping through one epoch.

Times microseconds per optimiser step for the single-chain and the concurrent-chain execution of the same pack;
run under ``rocprofv3 --kernel-trace`` with ``--streams N --epochs 1`` to look at kernel overlap.

    python bench/tail_regime.py [--slots 16] [--rows 200000] [--streams 1 2 4]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench as headline  # noqa: E402
from mitune.packed import PackedTrainable  # noqa: E402
from mitune.packed.engine_hip import HipPackEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--rows", type=int, default=200000)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--width", type=int, default=512)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    batches = [256, 512, 1024]
    for streams in args.streams:
        tr = PackedTrainable(train=train, val=val, max_epochs=4, engine="hip", capacity=args.slots, val_chunk=4096,
                             model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024, "streams": streams})
        eng = HipPackEngine(tr, dev, args.slots, 0)
        for i in range(args.slots):
            eng.load(i, i, {"hidden_sizes": [args.width] * 3, "batch_size": batches[i % 3], "learning_rate": 1e-3,
                            "optimizer": "adamw"})
        eng.train_epoch()                                   # warm-up: tables, graph capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            eng.train_epoch()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        steps = eng.n_train // 256
        print(f"slots {args.slots} streams {streams}: chains {len(eng._chains)}  {1e6 * el / (args.epochs * steps):8.1f} us/step "
              f"({steps} steps/epoch, ksplits {[ch.plan.ksplits for ch in eng._chains]})", flush=True)
        eng.close()


if __name__ == "__main__":
    main()
