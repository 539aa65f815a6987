#!/usr/bin/env python
"""The five target configurations as runnable scripts (each prints one JSON line on rank 0).

    python bench/run_config.py 1                      # 16-trial random+ASHA MLP search on the CPU (plumbing)
    python bench/run_config.py 2                      # = bench.py on one GPU (256 packed trials, 1M x 64)
    torchrun ... bench/run_config.py 3                # 1024 trials over 8 GPUs, RCCL metric all-gather
    [torchrun ...] bench/run_config.py 4 --rows N     # width-8192 MLP, every trial data-parallel over all ranks
    [torchrun ...] bench/run_config.py 5              # FT-Transformer, bf16, population-based training
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def _comm():
    from mitune.parallel.comm import LocalComm, TorchDistComm
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return TorchDistComm(backend="nccl" if torch.cuda.is_available() else "gloo")
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    return LocalComm(dev)


def config1(args):
    import mitune
    from mitune import tune
    from mitune.data import tabular_regression
    from mitune.packed import PackedTrainable
    from mitune.schedulers import ASHAScheduler
    train, val = tabular_regression(20000, 8, seed=1)
    tr = PackedTrainable(train=train, val=val, max_epochs=8, engine="torch", capacity=16, name="config1_mlp")
    t0 = time.perf_counter()
    an = mitune.run(tr, config={"width": tune.choice([16, 32, 64]), "num_layers": tune.choice([1, 2, 3]),
                                "learning_rate": tune.loguniform(1e-4, 1e-2), "batch_size": tune.choice([64, 128, 256]),
                                "weight_decay": tune.loguniform(1e-6, 1e-2)},
                    num_samples=16, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=8,
                                                            reduction_factor=2),
                    local_dir=args.local_dir, verbose=args.verbose, device=torch.device("cpu"))
    el = time.perf_counter() - t0
    return {"config": 1, "trials": len(an.trials), "elapsed_s": el, "trials_per_hour": len(an.trials) / el * 3600,
            "best_validation_mse": an.best_result["validation_mse"], "engine": an.stats["engine"]}


def config23(args, trials_per_gpu):
    """Configurations 2 and 3 are bench.py itself (one GPU: 256 packed trials; a node: 128 per GPU = 1024 on eight).  It
    runs in THIS process -- under torchrun every rank is already a member of the job, and a child process would try to
    join the rendezvous a second time with its parent's rank -- and prints its own JSON line on rank 0."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mitune_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    sys.argv = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--trials-per-gpu", str(trials_per_gpu),
                "--steps", str(args.steps), "--warmup", "1"] + (["--no-bf16"] if args.no_bf16 else [])
    bench.main()
    return None


def config4(args):
    """Width-8192 MLP on a [rows x 256] table: a learning-rate x batch-size search under ASHA in which every trial is
    data-parallel over all ranks (``mitune.run(dp_mlp_trainable, ...)``, trials one after another, every rank running the
    same replicated driver).  A trial reports after every ``--dp-steps`` optimiser steps ("epoch" of the scheduler);
    ``--dp-dtype both`` runs the search once in bf16 and once in exact f32."""
    import mitune
    from mitune import tune
    from mitune.parallel.dp import dp_mlp_trainable, shard_rows
    from mitune.schedulers import ASHAScheduler
    comm = _comm()
    dev, rank, world = comm.device, comm.rank, comm.world_size
    lo, hi = shard_rows(args.rows, rank, world)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    n_local = hi - lo
    w = torch.randn(256, generator=torch.Generator(device=dev).manual_seed(7), device=dev) / 16.0
    runs = []
    dtypes = ["bf16", "f32"] if args.dp_dtype == "both" else [args.dp_dtype]
    if dev.type != "cuda":
        dtypes = ["f32"]
    for dtype in dtypes:
        dt = torch.bfloat16 if dtype == "bf16" else torch.float32
        X = torch.empty(n_local, 256, device=dev, dtype=dt)
        y = torch.empty(n_local, device=dev, dtype=torch.float32)
        g.manual_seed(1000 + rank)
        for c0 in range(0, n_local, 1 << 20):        # generated in place, chunk by chunk
            c1 = min(n_local, c0 + (1 << 20))
            xc = torch.randn(c1 - c0, 256, generator=g, device=dev)
            y[c0:c1] = torch.tanh(xc @ w) + 0.1 * xc[:, 0] * xc[:, 1] + 0.05 * torch.randn(c1 - c0, generator=g, device=dev)
            X[c0:c1] = xc.to(dt)
        n_val = min(max(1024, n_local // 50), 262144)      # a quarter million held-out rows bound the cost of a report
        trainable = tune.with_parameters(dp_mlp_trainable, train=(X[n_val:], y[n_val:]), val=(X[:n_val], y[:n_val]), comm=comm,
                                         epochs=args.dp_epochs, seed=0)
        space = {"hidden_sizes": [args.width] * 3, "optimizer": "adamw", "dtype": dtype, "steps_per_epoch": args.dp_steps,
                 "weight_decay": 1e-4, "warmup_steps": args.dp_steps // 2,
                 "learning_rate": tune.loguniform(3e-5, 1e-3), "batch_size": tune.choice([8192, 16384])}
        comm.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        an = mitune.run(trainable, config=space, num_samples=args.trials, max_concurrent_trials=1, seed=0,
                        scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=args.dp_epochs, grace_period=1,
                                                reduction_factor=2),
                        metric="validation_loss", mode="min", local_dir=None, verbose=args.verbose if rank == 0 else 0)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        trials, flop, steps_total, train_s = [], 0.0, 0, 0.0
        for t in an.trials:
            bs, steps = int(t.config["batch_size"]), int(t.last_result.get("steps", 0))
            flop += 6.0 * (256 * args.width + 2 * args.width ** 2 + args.width) * bs * steps
            steps_total += steps
            trials.append({"lr": t.config["learning_rate"], "global_batch": bs, "reports": t.training_iteration, "steps": steps,
                           "train_loss": t.last_result.get("train_loss"), "validation_mse": t.last_result.get("validation_mse"),
                           "status": t.status})
        finite = [t["validation_mse"] for t in trials if t["validation_mse"] is not None and math.isfinite(t["validation_mse"])]
        runs.append({"dtype": dtype, "trials": len(trials), "elapsed_s": el, "trials_per_hour": len(trials) / el * 3600.0,
                     "optimiser_steps": steps_total, "useful_tflop": flop / 1e12, "tflops_whole_search": flop / el / 1e12,
                     "best_validation_mse": min(finite) if finite else None, "diverged": len(trials) - len(finite),
                     "budget": f"{args.dp_epochs} reports x {args.dp_steps} steps, ASHA reduction factor 2",
                     "per_trial": trials})
        del X, y, trainable
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    comm.close()
    return {"config": 4, "n_gpus": world, "rows": args.rows, "width": args.width, "searches": runs} if rank == 0 else None


def config5(args):
    import mitune
    from mitune import tune
    from mitune.data import tabular_regression
    from mitune.packed import PackedExecutor, PackedTrainable
    from mitune.schedulers import PopulationBasedTraining
    comm = _comm()
    dev = comm.device
    train, val = tabular_regression(args.rows5, 64, seed=5, device=dev)
    pop = args.population * comm.world_size
    tr = PackedTrainable(model="ft_transformer", train=train, val=val, max_epochs=args.steps5, engine=args.engine5,
                         dtype="bf16", capacity=args.population, name="config5_ft",
                         model_kwargs={"d_token": 192, "n_blocks": 3, "n_heads": 8})
    ex = PackedExecutor(tr, comm=comm, device=dev, capacity=args.population, seed=0)
    pbt = PopulationBasedTraining(metric="validation_loss", mode="min", perturbation_interval=2,
                                  hyperparam_mutations={"learning_rate": tune.loguniform(1e-5, 3e-3),
                                                        "weight_decay": tune.loguniform(1e-6, 1e-2)},
                                  quantile_fraction=0.25, seed=0)
    t0 = time.perf_counter()
    an = mitune.run(tr, config={"learning_rate": tune.loguniform(1e-5, 3e-3), "weight_decay": tune.loguniform(1e-6, 1e-2),
                                "batch_size": 1024, "optimizer": "adamw"},
                    num_samples=pop, scheduler=pbt, executor=ex, seed=0, verbose=args.verbose if comm.rank == 0 else 0)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    if an.best_result is None:
        raise RuntimeError(f"no finite result: {[t.last_result for t in an.trials][:3]} stats={an.stats}")
    out = {"config": 5, "n_gpus": comm.world_size, "population": pop, "epochs": args.steps5, "rows": args.rows5,
           "elapsed_s": el, "trials_per_hour": pop / el * 3600, "exploits": an.stats["exploits"],
           "best_validation_mse": an.best_result["validation_mse"], "dtype": "bf16",
           "engine": an.stats["engine"]}
    comm.close()
    return out if comm.rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", type=int, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--rows", type=int, default=100_000_000, help="config 4: total table rows")
    ap.add_argument("--width", type=int, default=8192)
    ap.add_argument("--trials", type=int, default=6, help="config 4: trials of the search")
    ap.add_argument("--dp-steps", type=int, default=25, help="config 4: optimiser steps between two reports of a trial")
    ap.add_argument("--dp-epochs", type=int, default=4, help="config 4: reports of a trial that is never stopped")
    ap.add_argument("--dp-dtype", default="both", choices=["bf16", "f32", "both"])
    ap.add_argument("--rows5", type=int, default=200_000)
    ap.add_argument("--steps5", type=int, default=6)
    ap.add_argument("--population", type=int, default=8)
    ap.add_argument("--engine5", default="auto", help="config 5 engine: auto (graph on a GPU) | graph | torch")
    ap.add_argument("--local-dir", default=None)
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--no-bf16", action="store_true", help="configs 2/3: skip bench.py's second (bf16) search")
    args = ap.parse_args()
    out = {1: config1, 2: lambda a: config23(a, 256), 3: lambda a: config23(a, 128), 4: config4, 5: config5}[args.config](args)
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
