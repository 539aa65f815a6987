#!/usr/bin/env python
"""Search-quality parity AT THE HEADLINE BUDGET: the 256-trial / 1M x 64 / ASHA(rf 4, max_t 16) search of bench.py run
three times with the same seed, data, configurations, initial weights and batch order --

  hip-f32    the packed exact-f32 engine (the headline),
  hip-bf16   the packed bf16 MFMA engine,
  eager-f32  the reference's execution model: one trial at a time, eager f32 PyTorch modules + torch.optim on the GPU
             (/root/reference/ray-tune-hpo-regression-sample.py:88-135), same ASHA decisions taken per round --

and reports for each: wall time, best validation MSE, the best trial, and the sets of trials that survived rung 1
(epoch 1) and rung 4, so that "equal-or-better best val MSE at a fixed HPO budget" is a measured statement.  Results
are appended to gpurun_out/parity_full.jsonl as each search ends (the eager one takes about a quarter of an hour).

    python bench/parity_full.py [--skip-eager] [--trials 256] [--rows 1000000] [--steps 16]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch         # noqa: E402

import bench as headline   # noqa: E402
import mitune              # noqa: E402
from mitune.packed import PackedTrainable            # noqa: E402
from mitune.packed.engine import TorchPackEngine     # noqa: E402
from mitune.packed.executor import PackedExecutor    # noqa: E402
from mitune.schedulers import ASHAScheduler          # noqa: E402


class ChattyTorchEngine(TorchPackEngine):
    """The eager engine, saying every few trials where it is (a silent quarter of an hour looks like a hang)."""
    t0 = time.time()
    rounds = 0

    def train_epoch(self):
        ChattyTorchEngine.rounds += 1
        slots = self.occupied()
        for k, slot in enumerate(slots):
            with self._cpu_threads(self.slots[slot]):
                self._train_slot_epoch(self.slots[slot])
            if k % 8 == 7 or k + 1 == len(slots):
                torch.cuda.synchronize()
                print(f"[eager-f32] round {ChattyTorchEngine.rounds}: {k + 1}/{len(slots)} trials trained, "
                      f"{time.time() - ChattyTorchEngine.t0:.0f} s", flush=True)


def search(kind, device, train, val, trials, steps, seed):
    kw = {"max_width": 512, "max_layers": 3, "max_batch": 1024}
    engine_name, dtype = {"hip-f32": ("hip", "f32"), "hip-bf16": ("hip", "bf16"), "eager-f32": ("torch", "f32")}[kind]
    tr = PackedTrainable(train=train, val=val, max_epochs=steps, engine=engine_name, dtype=dtype, capacity=trials,
                         val_chunk=4096, model_kwargs=kw, name="parity_full")
    engine = ChattyTorchEngine(tr, str(device), trials, seed) if kind == "eager-f32" else None
    ex = PackedExecutor(tr, device=device, capacity=trials, seed=seed, engine=engine)
    sched = ASHAScheduler(metric="validation_loss", mode="min", max_t=steps, grace_period=1, reduction_factor=4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    an = mitune.run(tr, config=headline.search_space(), num_samples=trials, scheduler=sched, executor=ex, seed=seed,
                    verbose=0, local_dir=None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    best = an.best_trial
    return {"engine": kind, "trials": trials, "steps": steps, "elapsed_s": el, "trials_per_hour": trials / el * 3600.0,
            "best_validation_mse": an.best_result["validation_mse"], "best_trial": best.index,
            "best_config": {k: v for k, v in best.config.items() if k.startswith("width") or k in ("batch_size", "learning_rate")},
            "trial_epochs": sum(t.training_iteration for t in an.trials),
            "past_rung1": sorted(t.index for t in an.trials if t.training_iteration > 1),
            "past_rung4": sorted(t.index for t in an.trials if t.training_iteration > 4),
            "epoch1_validation_mse": {t.index: t.results[0]["validation_mse"] for t in an.trials}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=256)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--skip-eager", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    train, val = headline.make_data(args.rows, 64, dev)
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_full.jsonl")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    runs = {}
    search("hip-f32", dev, train, val, 16, 1, args.seed + 1)            # warm-up (library load, allocator)
    for kind in ["hip-f32", "hip-bf16"] + ([] if args.skip_eager else ["eager-f32"]):
        if kind == "eager-f32":
            ChattyTorchEngine.t0 = time.time()
        r = search(kind, dev, train, val, args.trials, args.steps, args.seed)
        runs[kind] = r
        e1 = r.pop("epoch1_validation_mse")
        with open(out_path, "a") as f:
            f.write(json.dumps(r) + "\n")
        r["epoch1_validation_mse"] = e1
        print(f"[parity] {kind}: {r['elapsed_s']:.1f} s, best val MSE {r['best_validation_mse']:.6f} (trial {r['best_trial']}), "
              f"{len(r['past_rung1'])} past rung 1, {len(r['past_rung4'])} past rung 4", flush=True)
    ref = runs.get("eager-f32") or runs["hip-f32"]
    summary = {"reference": ref["engine"]}
    for kind, r in runs.items():
        if r is ref:
            continue
        both = [i for i in ref["epoch1_validation_mse"] if i in r["epoch1_validation_mse"]]
        rel = sorted(abs(r["epoch1_validation_mse"][i] - ref["epoch1_validation_mse"][i]) / max(ref["epoch1_validation_mse"][i], 1e-12)
                     for i in both)
        summary[kind] = {"same_best_trial": r["best_trial"] == ref["best_trial"],
                         "rung1_survivors_in_common": len(set(r["past_rung1"]) & set(ref["past_rung1"])),
                         "rung4_survivors_in_common": len(set(r["past_rung4"]) & set(ref["past_rung4"])),
                         "epoch1_rel_diff_median": rel[len(rel) // 2], "epoch1_rel_diff_p90": rel[int(0.9 * len(rel))],
                         "epoch1_rel_diff_max": rel[-1]}
    with open(out_path, "a") as f:
        f.write(json.dumps({"summary": summary}) + "\n")
    print(json.dumps({"summary": summary}))


if __name__ == "__main__":
    main()
