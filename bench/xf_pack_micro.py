#!/usr/bin/env python
"""One pack of the full search space on the program engine: host time to enqueue an epoch against the epoch's GPU time.

    python bench/xf_pack_micro.py [--slots 16] [--max-layers 16] [--epochs 2] [--seed 0] [--dtype bf16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import torch  # noqa: E402

from mitune import tune  # noqa: E402
from mitune.data.loaders import frames_to_loaders, synthetic_glucose_frame  # noqa: E402
from mitune.packed import PackedTrainable  # noqa: E402
from mitune.packed.engine_xf import XfPackEngine  # noqa: E402
from mitune.tune.space import generate_variants  # noqa: E402
from search_spaces import FULL_SPACE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=16)
    ap.add_argument("--max-layers", type=int, default=16)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16: bf16 products in the grouped GEMMs")
    args = ap.parse_args()
    tl, vl = frames_to_loaders(synthetic_glucose_frame(96 * 2000, seed=7))
    space = dict(FULL_SPACE)
    space["num_encoder_layers"] = tune.choice([n for n in (2, 4, 6, 8, 12, 16) if n <= args.max_layers])
    tr = PackedTrainable(model="transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors), max_epochs=5,
                         dtype=args.dtype, capacity=args.slots, name="xf_micro",
                         model_kwargs={"max_d_model": 512, "max_layers": args.max_layers, "max_dim_feedforward": 2048,
                                       "max_batch": 256})
    eng = XfPackEngine(tr, "cuda", args.slots, 0)
    for k, cfg in enumerate(generate_variants(space, args.slots, seed=args.seed)):
        eng.load(k, k, cfg)
        sp = eng.slots[k].spec
        print(f"slot {k:2d}: d {sp.d:3d} h {sp.heads:2d} dk {sp.dk:3d} L {sp.layers:2d} dff {sp.dff:4d} b {sp.opt.batch_size:3d} "
              f"{'relu' if sp.kind else 'soft'} {'dw' if sp.depthwise else 'ff'} {'shared' if sp.shared else ''} "
              f"{'fused' if sp.fused else 'WIDE'} steps {eng.n_train // min(sp.opt.batch_size, eng.n_train)}", flush=True)
    for ep in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng._refresh()
        t1 = time.perf_counter()
        eng.train_epoch()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        m = eng.evaluate()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"epoch {ep}: compile {t1 - t0:.3f} s, enqueue {t2 - t1:.3f} s, GPU tail after enqueue {t3 - t2:.3f} s, "
              f"evaluate {t4 - t3:.3f} s; chains (slots, ops) {[(len(c['ids']), c['n']) for c in eng._chains]}", flush=True)


if __name__ == "__main__":
    main()
