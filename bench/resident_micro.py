#!/usr/bin/env python
"""One training epoch of a late-round pack (the 16 survivors of the headline search, or --slots N random draws of its
search space) as grouped launches and as one resident launch: microseconds per optimiser step of the longest chain,
useful TFLOP/s, and whether the two paths left bit-identical weights.

    python bench/resident_micro.py [--rows 250000] [--slots 16] [--modes grouped,resident]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SURVIVORS = [(512, 256, 128, 256), (512, 128, 512, 256), (512, 512, 128, 512), (512, 512, 256, 256), (512, 128, 64, 256),
             (256, 512, 512, 256), (256, 256, 512, 256), (512, 64, 512, 256), (256, 512, 64, 512), (512, 128, 256, 256),
             (256, 64, 256, 256), (512, 512, 64, 256), (512, 128, 64, 256), (512, 64, 256, 256), (512, 256, 512, 512),
             (512, 64, 128, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=250_000)
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--modes", default="grouped,resident")
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args()
    import bench as headline
    from mitune.packed import PackedTrainable, engine_hip
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(args.rows, 64, dev)
    if args.slots == 16:
        cfgs = [{"hidden_sizes": [a, b, c], "batch_size": bs, "learning_rate": 1e-3, "optimizer": "adamw",
                 "weight_decay": 1e-4} for a, b, c, bs in SURVIVORS]
    else:
        g = torch.Generator().manual_seed(0)
        w = [64, 128, 256, 512]
        pick = lambda opts: opts[int(torch.randint(len(opts), (1,), generator=g))]      # noqa: E731
        cfgs = [{"hidden_sizes": [pick(w), pick(w), pick(w)], "batch_size": pick([256, 512, 1024]), "learning_rate": 1e-3,
                 "optimizer": "adamw", "weight_decay": 1e-4} for _ in range(args.slots)]
    n_tr = int(train[0].shape[0])
    flop = 0.0
    for c in cfgs:
        ws = [64] + c["hidden_sizes"] + [1]
        flop += 6.0 * sum(a * b for a, b in zip(ws[:-1], ws[1:])) * (n_tr // c["batch_size"]) * c["batch_size"]
    steps = n_tr // min(c["batch_size"] for c in cfgs)
    out, states = {}, {}
    for mode in args.modes.split(","):
        engine_hip.RESIDENT = "1" if mode == "resident" else "0"
        tr = PackedTrainable(train=train, val=val, max_epochs=args.epochs + 1, engine="hip", dtype="f32",
                             capacity=len(cfgs), val_chunk=4096,
                             model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
        eng = engine_hip.HipPackEngine(tr, dev, len(cfgs), 0)
        for i, c in enumerate(cfgs):
            eng.load(i, i, c)
        eng.train_epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            eng.train_epoch()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.epochs
        out[mode] = {"epoch_ms": dt * 1e3, "us_per_step": dt / steps * 1e6, "tflops": flop / dt / 1e12}
        states[mode] = eng.P.clone()
        prof = getattr(eng, "_res_prof", None)
        if mode == "resident" and prof is not None:      # ticks of the 100 MHz clock, all epochs (incl. warm-up), group 0 member 0
            t = prof.tolist()
            n = max(t[5], 1)
            out[mode]["group0_us_per_step"] = {k: round(v / n / 100.0, 2) for k, v in
                                               zip(("fwd", "head", "bwd", "opt", "barrier"), t[:5])}
            out[mode]["group0"] = eng._res["groups"][0]
        eng.close()
    if len(states) == 2:
        a, b = states.values()
        out["bit_identical"] = bool(torch.equal(a, b))
    print(json.dumps({"slots": len(cfgs), "rows": args.rows, "steps": steps, "sync": os.environ.get("MITUNE_RES_SYNC", "0"),
                      "tile": os.environ.get("MITUNE_RES_TILE", "auto"), "gmin": os.environ.get("MITUNE_RES_GMIN", "8"),
                      **out}))


if __name__ == "__main__":
    main()
