#!/usr/bin/env python
"""CPU-eager baseline for the packed-MLP search (the reference's execution model on its CPU path).

The reference trains one trial at a time with ordinary eager PyTorch modules and optimisers
(/root/reference/ray-tune-hpo-regression-sample.py:88-135).  It contains no MLP trainer, so this script is that
execution model applied to the benchmark's workload: it draws configurations from the SAME search space
and seed as ``bench.py``, trains each as a plain ``MLPRegressor`` + ``torch.optim.AdamW`` on the CPU for
one epoch over ``--rows`` training rows, and measures seconds per training row and per validation row.
Because an epoch's cost is linear in its row count, the cost of the full search (1M x 64 table, 800k
training rows, ASHA with reduction factor 4 and max_t epochs: 256 + 64*3 + 16*12 = 640 trial-epochs for
256 trials) is extrapolated from those rates and reported as trials/hour.

    python bench/baseline_cpu_eager.py --configs 12 --rows 40000 --threads 8
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench as headline                                       # noqa: E402  (search space + data generator)
from mitune.models.mlp import MLPRegressor, MLPTrialSpec     # noqa: E402
from mitune.tune import space as sp                           # noqa: E402


def asha_trial_epochs(n_trials: int, max_t: int, rf: int = 4, grace: int = 1) -> int:
    total, alive, t_prev, rung = 0, n_trials, 0, grace
    while rung < max_t:
        total += alive * (rung - t_prev)
        t_prev, alive, rung = rung, max(1, alive // rf), rung * rf
    return total + alive * (max_t - t_prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=12)
    ap.add_argument("--rows", type=int, default=40000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--trials", type=int, default=256)
    ap.add_argument("--max-t", type=int, default=16)
    ap.add_argument("--full-rows", type=int, default=1_000_000)
    ap.add_argument("--device", default="cpu", help="cpu (the reference's path here) or cuda (eager PyTorch, one trial "
                                                    "at a time on the MI355X: the reference's execution model on this GPU)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    dev = torch.device(args.device)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    (Xtr, ytr), (Xva, yva) = headline.make_data(int(args.rows / 0.8), 64, dev)
    cfgs = list(sp.generate_variants(headline.search_space(), args.configs, seed=0))
    t_train, t_val = [], []
    for cfg in cfgs:
        spec = MLPTrialSpec.from_config(cfg, 64)
        model = MLPRegressor(64, spec.hidden_sizes).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=spec.learning_rate, weight_decay=spec.weight_decay)
        crit = torch.nn.MSELoss()
        n = Xtr.shape[0] // spec.batch_size * spec.batch_size
        perm = torch.randperm(Xtr.shape[0], device=dev)[:n]
        sync()
        t0 = time.perf_counter()
        for i in range(0, n, spec.batch_size):
            idx = perm[i:i + spec.batch_size]
            opt.zero_grad()
            loss = crit(model(Xtr[idx]), ytr[idx])
            loss.backward()
            opt.step()
        sync()
        t1 = time.perf_counter()
        with torch.no_grad():
            for c0 in range(0, Xva.shape[0], 4096):
                crit(model(Xva[c0:c0 + 4096]), yva[c0:c0 + 4096])
        sync()
        t2 = time.perf_counter()
        t_train.append((t1 - t0) / n)
        t_val.append((t2 - t1) / Xva.shape[0])
        print(f"  {spec.hidden_sizes} bs={spec.batch_size}: {1e6 * t_train[-1]:.2f} us/train-row, "
              f"{1e6 * t_val[-1]:.2f} us/val-row", flush=True)
    train_rows, val_rows = int(0.8 * args.full_rows), args.full_rows - int(0.8 * args.full_rows)
    per_epoch = float(np.mean(t_train)) * train_rows + float(np.mean(t_val)) * val_rows
    epochs = asha_trial_epochs(args.trials, args.max_t)
    total_s = per_epoch * epochs
    out = {"metric": "trials_completed_per_hour", "engine": f"pytorch-eager-{dev.type}", "threads": args.threads,
           "configs_timed": args.configs, "rows_timed": int(Xtr.shape[0]),
           "us_per_train_row": 1e6 * float(np.mean(t_train)), "us_per_val_row": 1e6 * float(np.mean(t_val)),
           "seconds_per_trial_epoch_800k": per_epoch, "trial_epochs": epochs, "search_seconds_sequential": total_s,
           "trials_per_hour_sequential": args.trials / total_s * 3600.0,
           "trials_per_hour_4_concurrent_2cpu_upper_bound": None}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
