#!/usr/bin/env python
"""Cost of moving one trial of the headline pack between engines: slab export + import on one GPU (the transfer in
between is one broadcast of the same buffer over the job's communicator; timed here on a one-rank RCCL group).

    python bench/move_cost.py [--reps 20]
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
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(50000, 64, dev)
    tr = PackedTrainable(train=train, val=val, max_epochs=4, engine="hip", capacity=32, val_chunk=4096,
                         model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
    a, b = HipPackEngine(tr, dev, 32, 0), HipPackEngine(tr, dev, 32, 0)
    cfg = {"hidden_sizes": [512, 512, 512], "batch_size": 256, "learning_rate": 1e-3, "optimizer": "adamw"}
    for i in range(args.reps):
        a.load(i, i, cfg)
    a.train_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.reps):
        raw = a.export_raw(i)
        a.evict(i)
        b.import_raw(i, i, cfg, raw)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"export_raw + evict + import_raw: {1e3 * el / args.reps:.3f} ms per trial "
          f"({raw.numel() * 4 / 1e6:.1f} MB slab)", flush=True)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    from mitune.parallel.comm import TorchDistComm
    comm = TorchDistComm(backend="nccl", rank=0, world_size=1)
    comm.broadcast(raw, src=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        comm.broadcast(raw, src=0)
    torch.cuda.synchronize()
    print(f"RCCL broadcast call on a one-rank group: {1e3 * (time.perf_counter() - t0) / args.reps:.3f} ms", flush=True)
    b.train_epoch()
    print("destination pack trains on:", len(b.evaluate()), "trials", flush=True)
    comm.close()


if __name__ == "__main__":
    main()
