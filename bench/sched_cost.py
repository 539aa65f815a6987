#!/usr/bin/env python
"""Scheduling-only cost of a packed search: N trials (default 2048) under ASHA with an engine that trains nothing -- what
every rank of a replicated-scheduler run pays on the host besides its own training.  usage: python bench/sched_cost.py [N]"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import numpy as np, torch
import mitune
from mitune import tune
from mitune.packed import PackedTrainable
from mitune.packed.engine import PackEngine
from mitune.packed.executor import PackedExecutor
from mitune.schedulers import ASHAScheduler
from search_spaces import MLP_SPACE

class NullEngine(PackEngine):
    name = "null"
    def __init__(self, capacity):
        self.capacity = capacity; self.slots = {}; self.rng = np.random.default_rng(0)
    def load(self, slot, trial_index, config, state=None): self.slots[slot] = [trial_index, 0]
    def evict(self, slot): self.slots.pop(slot, None)
    def occupied(self): return sorted(self.slots)
    def train_epoch(self):
        for s in self.slots.values(): s[1] += 1
    def evaluate(self):
        return {k: {"train_loss": 1.0, "validation_loss": float(self.rng.random()) / v[1], "validation_mse": 0.1, "validation_mape": 1.0} for k, v in self.slots.items()}

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
X = torch.zeros(64, 64); y = torch.zeros(64)
tr = PackedTrainable(train=(X, y), val=(X, y), max_epochs=16, engine="torch", capacity=N)
def go():
    ex = PackedExecutor(tr, engine=NullEngine(N), device=torch.device("cpu"), capacity=N, seed=0)
    t0 = time.perf_counter()
    an = mitune.run(tr, config=MLP_SPACE, num_samples=N, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=16, reduction_factor=4),
                    executor=ex, seed=0, verbose=0, local_dir=None)
    return time.perf_counter() - t0, an
el, an = go()
print("scheduling-only", N, "trials:", round(el, 3), "s; results", an.stats.get("num_results"))
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
