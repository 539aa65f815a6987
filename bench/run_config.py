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
    """Width-8192 MLP on a [rows x 256] table, each trial data-parallel over all ranks."""
    import mitune
    from mitune import tune
    from mitune.parallel.dp import DPTrainer, shard_rows
    comm = _comm()
    dev, rank, world = comm.device, comm.rank, comm.world_size
    lo, hi = shard_rows(args.rows, rank, world)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    n_local = hi - lo
    dt = torch.bfloat16 if dev.type == "cuda" else torch.float32
    X = torch.empty(n_local, 256, device=dev, dtype=dt)
    y = torch.empty(n_local, device=dev, dtype=torch.float32)
    w = torch.randn(256, generator=torch.Generator(device=dev).manual_seed(7), device=dev) / 16.0
    for c0 in range(0, n_local, 1 << 20):        # generated in place, chunk by chunk: the shard never exists in f32
        c1 = min(n_local, c0 + (1 << 20))
        xc = torch.randn(c1 - c0, 256, generator=g, device=dev)
        y[c0:c1] = torch.tanh(xc @ w) + 0.1 * xc[:, 0] * xc[:, 1] + 0.05 * torch.randn(c1 - c0, generator=g, device=dev)
        X[c0:c1] = xc.to(dt)
    n_val = max(1024, n_local // 50)
    Xtr, ytr, Xva, yva = X[n_val:], y[n_val:], X[:n_val], y[:n_val]
    results = []
    for i, (lr, bs) in enumerate([(3e-4, 8192), (1e-3, 16384)][: args.trials]):
        cfg = {"hidden_sizes": [args.width] * 3, "batch_size": bs, "learning_rate": lr, "optimizer": "adamw"}
        tr = DPTrainer(cfg, Xtr, ytr, Xva, yva, comm=comm, seed=0, trial_index=i)
        steps = min(args.dp_steps, tr.steps_per_epoch)
        full_epoch = tr.steps_per_epoch
        tr.steps_per_epoch = steps                 # a timed slice of the epoch (the same number of steps on every rank)
        comm.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loss = tr.train_epoch()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        tr.steps_per_epoch = full_epoch
        m = tr.evaluate()
        flop = 6.0 * (256 * args.width + 2 * args.width ** 2 + args.width) * bs * steps
        results.append({"lr": lr, "global_batch": bs, "steps": steps, "ms_per_step": el / steps * 1e3,
                        "samples_per_s": bs * steps / el, "tflops": flop / el / 1e12, "train_loss": loss,
                        "validation_mse": m["validation_mse"], "engine": "hip-dp" if tr.engine is not None else "torch",
                        "allreduce_gb_per_step": tr.reducer.bytes_reduced / 1e9 / max(steps, 1),
                        "epoch_seconds_full_table": args.rows * 0.98 / (bs * steps / el)})
    comm.close()
    return {"config": 4, "n_gpus": world, "rows": args.rows, "width": args.width, "trials": results} if rank == 0 else None


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
    ap.add_argument("--trials", type=int, default=2)
    ap.add_argument("--dp-steps", type=int, default=50)
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
