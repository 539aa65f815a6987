"""Epoch time of an exact-f32 MLP pack under the two chain policies (mixed deal / one chain per batch size).

    python bench/chain_probe.py [--slots 16 64 256] [--rows 200000]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench as headline  # noqa: E402
from mitune.options import OPTIONS  # noqa: E402
from mitune.packed import PackedTrainable, engine_hip  # noqa: E402
from mitune.tune import space as sp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, nargs="+", default=[16, 64, 256])
    ap.add_argument("--rows", type=int, default=250000)
    ap.add_argument("--dtype", default="f32")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    for slots in args.slots:
        for by_batch in (False, True):
            OPTIONS.chains_by_batch = by_batch
            tr = PackedTrainable(train=train, val=val, max_epochs=3, engine="hip", dtype=args.dtype, capacity=slots,
                                 val_chunk=4096, model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
            eng = engine_hip.HipPackEngine(tr, dev, slots, 0)
            for i, cfg in enumerate(sp.generate_variants(headline.search_space(), slots, seed=0)):
                eng.load(i, i, cfg)
            eng.train_epoch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                eng.train_epoch()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            chains = [(len(ch.ids), ch.steps) for ch in eng._chains]
            mse = sorted(round(v["validation_mse"], 9) for v in eng.evaluate().values())[:2]
            print(f"slots {slots:4d} by_batch {int(by_batch)}: epoch {dt * 1e3:8.1f} ms  chains (slots, steps) {chains}  best mse {mse}",
                  flush=True)
            eng.close()
            del eng


if __name__ == "__main__":
    main()
