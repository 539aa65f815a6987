#!/usr/bin/env python
"""The reference's full workflow on this framework: the configurable Transformer over the 20-key search space.

Search space, model family, optimisers / losses / LR schedule / clipping and the ASHA-on-validation_mape set-up are
those of /root/reference/ray-tune-hpo-regression.py:266-400,465-480; the wearable-sensor tables are private, so the
data is the synthetic stand-in of the same shape (96-step windows of the engineered feature columns -> next glucose
value).  ``--packed K`` runs the search through the packed executor (K trials resident on the GPU, each with its own
captured step); without it every trial is a function trainable, one at a time, as in the reference.

    python bench/full_workload.py [--num-samples 32] [--epochs 5] [--packed 16] [--max-layers 6] [--bayes]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

import torch  # noqa: E402

from mitune import tune                                          # noqa: E402
from mitune.data.loaders import frames_to_loaders, synthetic_glucose_frame   # noqa: E402
from mitune.schedulers import ASHAScheduler                      # noqa: E402
from mitune.train.functions import train_transformer_model       # noqa: E402
from search_spaces import FULL_SPACE                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-samples", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--packed", type=int, default=0)
    ap.add_argument("--max-layers", type=int, default=6, help="cap on num_encoder_layers")
    ap.add_argument("--max-width", type=int, default=512, help="cap on d_model")
    ap.add_argument("--bayes", action="store_true", help="GP-UCB search as in the reference (default: random search)")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--warm", type=int, default=2)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="packed executor: f32, or bf16 autocast")
    ap.add_argument("--windows", type=int, default=2000, help="96-step windows in the synthetic table (70 % train)")
    args = ap.parse_args()
    tl, vl = frames_to_loaders(synthetic_glucose_frame(96 * args.windows, seed=7))
    space = dict(FULL_SPACE, device=args.device)
    space["num_encoder_layers"] = tune.choice([n for n in (2, 4, 6, 8, 12, 16) if n <= args.max_layers])
    space["d_model"] = tune.choice([d for d in (64, 128, 192, 256, 320, 512) if d <= args.max_width])

    def search(n, seed):
        sched = ASHAScheduler(metric="validation_mape", mode="min", max_t=args.epochs)
        alg = None
        if args.bayes:
            from mitune.search import BayesOptSearch
            alg = BayesOptSearch(metric="validation_mape", mode="min", random_search_steps=10)
        if args.packed:
            from mitune.packed import PackedTrainable
            tr = PackedTrainable(model="transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                                 max_epochs=args.epochs, dtype=args.dtype, capacity=args.packed, name="full_packed")
            return tune.run(tr, config=space, num_samples=n, scheduler=sched, search_alg=alg, local_dir=None, verbose=0,
                            seed=seed, metric="validation_mape", mode="min", device=torch.device(args.device))
        return tune.run(tune.with_parameters(train_transformer_model, data_loader=tl, val_loader=vl, epochs=args.epochs),
                        config=space, num_samples=n, scheduler=sched, search_alg=alg,
                        resources_per_trial={"cpu": 1, "gpu": 1} if args.device == "cuda" else {"cpu": 1},
                        local_dir=None, verbose=0, seed=seed, metric="validation_mape", mode="min")

    if args.warm:
        search(args.warm, args.seed + 1)
    if args.device == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    an = search(args.num_samples, args.seed)
    if args.device == "cuda":
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    X = tl.dataset.tensors[0]
    print(json.dumps({"workload": "full-transformer-search", "device": args.device, "trials": len(an.trials),
                      "packed_slots": args.packed, "dtype": args.dtype if args.packed else "f32", "bayes": args.bayes, "epochs": args.epochs,
                      "data": f"synthetic {list(X.shape)}", "elapsed_s": el, "trials_per_hour": len(an.trials) / el * 3600.0,
                      "best_validation_mape": an.best_result["validation_mape"],
                      "epochs_run": sum(t.training_iteration for t in an.trials),
                      "errors": sum(1 for t in an.trials if t.status == "ERROR"),
                      "stats": {k: v for k, v in an.stats.items() if isinstance(v, (int, float, str))}}))


if __name__ == "__main__":
    main()
