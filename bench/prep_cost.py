#!/usr/bin/env python
"""Host time of preparing one sample-Transformer trial on the GPU, stage by stage: module construction on the CPU, upload,
flat parameter slab, device-state optimiser, warm-up + capture of the training step.  usage: python bench/prep_cost.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitune.data import create_dummy_data
from mitune.models.transformer import SimpleTransformerModel
from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
import mitune.train.graphed as G
dev = torch.device("cuda")
tl, vl = create_dummy_data()
X, Y = (t.to(dev) for t in tl.dataset.tensors)
def once(d, h, L):
    t = [time.perf_counter()]
    model = SimpleTransformerModel(10, d, h, L, 0.2); t.append(time.perf_counter())
    model = model.to(dev); t.append(time.perf_counter())
    flat = FlatParams(model, grad_slab=False); t.append(time.perf_counter())
    opt = FlatOptimizer(flat, "adamw", 1e-3, 1e-4); t.append(time.perf_counter())
    tr = GraphedTrainer(model, torch.nn.MSELoss(), X, Y, 32, opt, shuffle=False, drop_last=True); t.append(time.perf_counter())
    return [round((b - a) * 1e3, 2) for a, b in zip(t, t[1:])]
for _ in range(3): once(64, 4, 2)
torch.cuda.synchronize()
import numpy as np
for cfg in ((32, 2, 1), (64, 4, 2), (128, 4, 3)):
    rows = np.array([once(*cfg) for _ in range(8)])
    print(cfg, "ms: build-cpu, to(dev), FlatParams, FlatOptimizer, trainer(warm+capture) =", rows.mean(0).round(2), "total", rows.sum(1).mean().round(2), flush=True)
