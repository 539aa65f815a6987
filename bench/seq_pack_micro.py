"""Step time of a pack of sample-space Transformer trials: grouped launches ('seq') against one captured step per slot
replayed over HIP streams ('graph').

    python bench/seq_pack_micro.py [--slots 1,4,16,64] [--epochs 2] [--engines seq,graph]

Every slot holds a trial drawn from the reference's sample search space
(/root/reference/ray-tune-hpo-regression-sample.py:140-147); data has the sample script's shape (800 x 50 x 10 train,
200 validation, batch 32 -> 25 steps per epoch).  Prints microseconds per pack step and per slot step, and the useful
f32 FLOP rate of the dense layers.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def flop_per_step(cfg, batch=32, S=50, F=10):
    d, L = cfg["d_model"], cfg["num_layers"]
    m = batch * S
    dense = 2 * m * (F * d) + L * 2 * m * (3 * d * d + d * d + 8 * d * d)
    attn = L * 4 * batch * S * S * d
    return 3 * (dense + attn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", default="1,4,16,64")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--engines", default="seq,graph")
    ap.add_argument("--fixed", default="", help="d_model,heads,layers for every slot instead of random draws")
    args = ap.parse_args()
    import random

    from mitune.packed import PackedTrainable
    from mitune.packed.executor import make_engine
    g = torch.Generator().manual_seed(0)
    X, y = torch.rand(1000, 50, 10, generator=g), torch.rand(1000, generator=g)
    rng = random.Random(0)
    for n in [int(v) for v in args.slots.split(",")]:
        cfgs = []
        for _ in range(n):
            if args.fixed:
                d, h, l = [int(v) for v in args.fixed.split(",")]
            else:
                d, h, l = rng.choice([32, 64, 128]), rng.choice([2, 4]), rng.choice([1, 2, 3])
            cfgs.append({"d_model": d, "num_heads": h, "num_layers": l, "dropout": rng.uniform(0.1, 0.5),
                         "learning_rate": 1e-3, "weight_decay": 1e-4})
        flop = sum(flop_per_step(c) for c in cfgs)
        for kind in args.engines.split(","):
            tr = PackedTrainable(model="simple_transformer", train=(X[:800], y[:800]), val=(X[800:], y[800:]),
                                 max_epochs=5, dtype="f32", engine=kind, capacity=n,
                                 model_kwargs={"batch_size": 32, "optimizer": "adamw"})
            eng = make_engine(tr, torch.device("cuda"), n, 0)
            for i, c in enumerate(cfgs):
                eng.load(i, i, c)
            eng.train_epoch()
            eng.evaluate()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.epochs):
                eng.train_epoch()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (args.epochs * 25)
            t1 = time.perf_counter()
            m = eng.evaluate()
            torch.cuda.synchronize()
            te = time.perf_counter() - t1
            print(f"{kind:6s} slots {n:4d}  {dt * 1e6:9.1f} us / pack step  {dt * 1e6 / n:8.1f} us / slot step  "
                  f"{flop / dt / 1e12:6.2f} TFLOP/s useful  eval {te * 1e3:7.2f} ms  "
                  f"val mse[0] {m[0]['validation_mse']:.4f}", flush=True)
            if hasattr(eng, "close"):
                eng.close()
            del eng
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
