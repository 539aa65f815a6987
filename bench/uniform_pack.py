#!/usr/bin/env python
"""A search over learning rate and weight decay only: every trial the same 64-512-512-512-1 MLP (the common case of a
fixed architecture), 256 trials packed on one GPU, a few epochs without early stopping.  Prints seconds per epoch and
useful TFLOP/s for the bf16 engine with and without 256-tiles, and for the exact-f32 engine.

    python bench/uniform_pack.py [--rows 200000] [--epochs 2] [--slots 256]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=200_000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--batch", type=int, default=512)
    args = ap.parse_args()
    import bench as headline
    from mitune.options import OPTIONS
    from mitune.packed import PackedTrainable, engine_hip
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    g = torch.Generator().manual_seed(0)
    cfgs = [{"hidden_sizes": [512, 512, 512], "batch_size": args.batch, "optimizer": "adamw",
             "learning_rate": float(10 ** (-4 + 2 * torch.rand(1, generator=g).item())),
             "weight_decay": float(10 ** (-6 + 4 * torch.rand(1, generator=g).item()))} for _ in range(args.slots)]
    n_tr = int(train[0].shape[0])
    macs = 64 * 512 + 2 * 512 * 512 + 512
    flop = 6.0 * macs * (n_tr // args.batch) * args.batch * args.slots
    out = {}
    for name, dtype, big in (("bf16_128_tiles", "bf16", 0), ("bf16_256_tiles", "bf16", 256), ("f32", "f32", 0)):
        OPTIONS.big_min_tiles = big
        tr = PackedTrainable(train=train, val=val, max_epochs=args.epochs + 1, engine="hip", dtype=dtype,
                             capacity=args.slots, val_chunk=4096,
                             model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
        eng = engine_hip.HipPackEngine(tr, dev, args.slots, 0)
        for i, c in enumerate(cfgs):
            eng.load(i, i, c)
        eng.train_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            eng.train_epoch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.epochs
        m = eng.evaluate()
        out[name] = {"epoch_s": round(dt, 4), "us_per_step": round(dt / (n_tr // args.batch) * 1e6, 1),
                     "useful_tflops": round(flop / dt / 1e12, 1),
                     "mean_val_mse": round(sum(v["validation_mse"] for v in m.values()) / len(m), 6)}
        eng.close()
        del eng
        torch.cuda.empty_cache()
    print(json.dumps({"slots": args.slots, "rows": args.rows, "batch": args.batch, **out}))


if __name__ == "__main__":
    main()
