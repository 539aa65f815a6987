#!/usr/bin/env python
"""The reference's own sample workload on this framework: N trials of the small-Transformer random search.

Same data shape ([1000, 50, 10] -> [1000], 80/20 split), model family, search space, batch size 32 and
5 epochs x 25 steps per trial as /root/reference/ray-tune-hpo-regression-sample.py:28-55,60-83,110,140-147
(10 trials there, sequential CPU: 57.7 s = 624 trials/hour on an 8-vCPU box).  Differences, all on purpose:
one report per epoch (so ASHA is live), a learnable target unless --noise-target, and on a GPU the
graph-replayed training path.  ``--concurrent`` trials share the GPU (each on its own stream).  ``--packed K`` runs
the search through the packed executor instead: K trials resident at a time, each with its own captured step,
replayed round-robin over a few HIP streams by one host thread (``packed/engine_graph.py``).

    python bench/sample_workload.py [--num-samples 10] [--concurrent 1] [--device cuda|cpu] [--no-asha] [--packed K]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

import torch  # noqa: E402

from mitune import tune                               # noqa: E402
from mitune.data import create_dummy_data             # noqa: E402
from mitune.schedulers import ASHAScheduler, FIFOScheduler  # noqa: E402
from mitune.train.functions import train_dummy_model  # noqa: E402
from search_spaces import SAMPLE_SPACE                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-samples", type=int, default=10)
    ap.add_argument("--concurrent", type=int, default=1)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--no-asha", action="store_true", help="every trial runs all 5 epochs, as in the reference")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--noise-target", action="store_true")
    ap.add_argument("--warm", type=int, default=1, help="untimed trials first (library load, allocator warm-up)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--packed", type=int, default=0, help="trial slots of the packed executor (0: one function trial at a time)")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--engine", default="auto", help="packed engine: auto (grouped launches, 'seq') | graph (one captured step per slot) | torch")
    args = ap.parse_args()
    tl, vl = create_dummy_data(learnable=not args.noise_target)
    space = dict(SAMPLE_SPACE, device=args.device, hip_graph=not args.no_graph)
    res = {"cpu": 1, "gpu": 1.0 / args.concurrent} if args.device == "cuda" else {"cpu": 1}

    def packed_search(n, seed):
        from mitune.packed import PackedTrainable
        Xt, yt = tl.dataset.tensors
        Xv, yv = vl.dataset.tensors
        tr = PackedTrainable(model="simple_transformer", train=(Xt, yt), val=(Xv, yv), max_epochs=5, dtype="f32",
                             capacity=args.packed, name="sample_packed", engine=args.engine,
                             model_kwargs={"batch_size": 32, "optimizer": "adamw", "streams": args.streams})
        sched = FIFOScheduler() if args.no_asha else ASHAScheduler(metric="validation_loss", mode="min", max_t=5)
        return tune.run(tr, config=SAMPLE_SPACE, num_samples=n, scheduler=sched, local_dir=None, verbose=0, seed=seed,
                        metric="validation_loss", mode="min", device=torch.device(args.device))

    def search(n, seed):
        if args.packed:
            return packed_search(n, seed)
        sched = FIFOScheduler() if args.no_asha else ASHAScheduler(metric="validation_loss", mode="min", max_t=5)
        return tune.run(tune.with_parameters(train_dummy_model, data_loader=tl, val_loader=vl, epochs=5), config=space,
                        num_samples=n, scheduler=sched, resources_per_trial=res, max_concurrent_trials=args.concurrent,
                        local_dir=None, verbose=0, seed=seed, metric="validation_loss", mode="min")

    if args.warm:
        search(args.warm, args.seed + 1)
    if args.device == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    an = search(args.num_samples, args.seed)
    if args.device == "cuda":
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = sorted(t.last_result.get("time_total_s", 0.0) for t in an.trials)
    print(json.dumps({"workload": "sample-transformer-search", "device": args.device, "trials": len(an.trials),
                      "concurrent": args.concurrent, "packed_slots": args.packed, "streams": args.streams, "asha": not args.no_asha, "graphed": not args.no_graph,
                      "elapsed_s": el, "trials_per_hour": len(an.trials) / el * 3600.0,
                      "best_validation_loss": an.best_result["validation_loss"],
                      "trial_seconds_min_med_max": [per[0], per[len(per) // 2], per[-1]],
                      "epochs_run": sum(t.training_iteration for t in an.trials),
                      "executor": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in an.stats.items()
                                   if k in ("train_s", "eval_s", "comm_s", "epochs", "slot_epochs", "prepared_ahead", "engine", "stream_window")}}))


if __name__ == "__main__":
    main()
