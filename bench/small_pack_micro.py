#!/usr/bin/env python
"""Microseconds per optimiser step of a small exact-f32 MLP pack: the 16 survivors of the headline search (their widths
and batch sizes as the seed-0 search of bench.py produces them) trained by tile launches and by the row-block step.

    python bench/small_pack_micro.py [--rows 250000] [--forms 0,1] [--epochs 2]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

SURVIVORS = [([512, 256, 128], 256), ([512, 128, 512], 256), ([512, 512, 128], 512), ([512, 512, 256], 256),
             ([512, 128, 64], 256), ([256, 512, 512], 256), ([256, 256, 512], 256), ([512, 64, 512], 256),
             ([256, 512, 64], 512), ([512, 128, 256], 256), ([256, 64, 256], 256), ([512, 512, 64], 256),
             ([512, 128, 64], 256), ([512, 64, 256], 256), ([512, 256, 512], 512), ([512, 64, 128], 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=250_000)
    ap.add_argument("--forms", default="0,1")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--slots", type=int, default=16)
    args = ap.parse_args()
    import bench as headline
    from mitune.packed import PackedTrainable, engine_hip
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    out = {}
    for form in args.forms.split(","):
        engine_hip.ROWBLOCK = form
        tr = PackedTrainable(train=train, val=val, max_epochs=args.epochs, engine="hip", dtype="f32", capacity=args.slots,
                             val_chunk=4096, model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
        eng = engine_hip.HipPackEngine(tr, device=dev, capacity=args.slots, seed=0)
        for slot, (widths, bs) in enumerate(SURVIVORS[:args.slots]):
            eng.load(slot, slot, {"hidden_sizes": widths, "batch_size": bs, "learning_rate": 2e-3, "weight_decay": 1e-4,
                                  "optimizer": "adamw", "loss_function": "mse"})
        eng.train_epoch()                       # warm-up: tables, first touch
        torch.cuda.synchronize()
        steps = max(eng._round_steps.values())
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            eng.train_epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ev = eng.evaluate()
        out[form] = {"us_per_step": dt / (args.epochs * steps) * 1e6, "steps_per_epoch": steps,
                     "rb_blocks": [int(ch.plan.rb_nblocks) for ch in eng._chains],
                     "val_mse_slot0": ev[0]["validation_mse"], "val_mse_slot3": ev[3]["validation_mse"] if 3 in ev else None}
        eng.close()
        del eng
    print(json.dumps({"slots": args.slots, "rows": args.rows, "forms": out}))


if __name__ == "__main__":
    main()
