#!/usr/bin/env python
"""One sample-Transformer trial (d_model 64, 4 heads, 2 layers, batch 32 x 50 tokens) stepping through eight epochs of a
captured training step: the workload behind the per-step kernel inventories in ``profiles/``.

    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 bench/step_inventory.py
    python3 bench/step_inventory.py --summarise OUT/.../*kernel_stats.csv      # kernels per step, busy time per step
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EPOCHS, STEPS_PER_EPOCH = 8, 25


def summarise(path, steps=EPOCHS * STEPS_PER_EPOCH + 1):      # + the eager warm-up step before the capture
    rows = list(csv.DictReader(open(path)))
    tot = n = 0.0
    for r in rows:
        c = int(r["Calls"])
        if c < 100:
            continue
        per = c / steps
        tot += float(r["TotalDurationNs"]) / steps / 1e3
        n += per
        print(f"{per:6.1f}/step  avg {float(r['AverageNs']) / 1e3:6.1f} us  {r['Name'][:120]}")
    print("kernels/step", round(n, 1), "busy us/step", round(tot, 1))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2])
    import torch
    from mitune.data import create_dummy_data
    from mitune.models.transformer import SimpleTransformerModel
    from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
    dev = torch.device("cuda")
    tl, _ = create_dummy_data()
    X, Y = tl.dataset.tensors
    model = SimpleTransformerModel(10, 64, 4, 2, 0.2).to(dev)
    opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adamw", 1e-3, 1e-4)
    tr = GraphedTrainer(model, torch.nn.MSELoss(), X.to(dev), Y.to(dev), 32, opt)
    for _ in range(EPOCHS):
        tr.train_epoch()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
