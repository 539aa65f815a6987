#!/usr/bin/env python
"""Search-quality parity at a fixed budget: f32 eager engine on the CPU vs the bf16 HIP pack engine on the GPU.

Both runs use the same data, search space, seed, ASHA settings and per-trial initialisation / shuffling
streams, so every trial index is the same configuration with the same starting weights in both; what
differs is the arithmetic (f32 eager PyTorch ops vs bf16 MFMA GEMMs with f32 accumulation and f32 master
weights).  Prints the best validation MSE of each search, whether both picked the same best trial, and
the distribution of per-trial relative differences of the final validation MSE.

    python bench/mse_parity.py [--trials 48] [--rows 60000] [--epochs 4]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np   # noqa: E402
import torch         # noqa: E402

import bench as headline   # noqa: E402
import mitune              # noqa: E402
from mitune.packed import PackedTrainable            # noqa: E402
from mitune.packed.executor import PackedExecutor    # noqa: E402
from mitune.schedulers import ASHAScheduler          # noqa: E402


def search(engine, device, train, val, trials, epochs, seed):
    kw = {"max_width": 512, "max_layers": 3, "max_batch": 1024}
    tr = PackedTrainable(train=train, val=val, max_epochs=epochs, engine=engine, dtype="bf16" if engine == "hip" else "f32",
                         capacity=trials, val_chunk=4096, model_kwargs=kw, name="parity")
    ex = PackedExecutor(tr, device=device, capacity=trials, seed=seed)
    sched = ASHAScheduler(metric="validation_loss", mode="min", max_t=epochs, grace_period=1, reduction_factor=4)
    return mitune.run(tr, config=headline.search_space(), num_samples=trials, scheduler=sched, executor=ex, seed=seed,
                      verbose=0, local_dir=None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=48)
    ap.add_argument("--rows", type=int, default=60000)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    cpu, gpu = torch.device("cpu"), torch.device("cuda", 0)
    (Xtr, ytr), (Xva, yva) = headline.make_data(args.rows, 64, gpu)
    a_gpu = search("hip", gpu, (Xtr, ytr), (Xva, yva), args.trials, args.epochs, args.seed)
    a_cpu = search("torch", cpu, (Xtr.cpu(), ytr.cpu()), (Xva.cpu(), yva.cpu()), args.trials, args.epochs, args.seed)
    first_g = {t.index: t.results[0]["validation_mse"] for t in a_gpu.trials}
    first_c = {t.index: t.results[0]["validation_mse"] for t in a_cpu.trials}
    rel = np.array([abs(first_g[i] - first_c[i]) / max(first_c[i], 1e-12) for i in sorted(first_c)
                    if np.isfinite(first_c[i]) and np.isfinite(first_g[i])])
    out = {
        "trials": args.trials, "rows": args.rows, "epochs": args.epochs,
        "best_val_mse_cpu_f32_eager": a_cpu.best_result["validation_mse"],
        "best_val_mse_gpu_bf16_hip": a_gpu.best_result["validation_mse"],
        "best_trial_cpu": a_cpu.best_trial.index, "best_trial_gpu": a_gpu.best_trial.index,
        "epoch1_rel_diff_median": float(np.median(rel)), "epoch1_rel_diff_p90": float(np.percentile(rel, 90)),
        "epoch1_rel_diff_max": float(rel.max()),
        "survivors_cpu": sum(t.training_iteration == args.epochs for t in a_cpu.trials),
        "survivors_gpu": sum(t.training_iteration == args.epochs for t in a_gpu.trials),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
