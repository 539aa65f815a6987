#!/usr/bin/env python
"""What moving trials between GPUs buys on a node: the headline search's placement, replayed without training.

N ranks (threads wired through the in-process communicator) run the real executor, runner and ASHA scheduler over a
stand-in engine whose losses are a fixed pseudo-random function of (trial, epoch), so that early stopping thins the
ranks unevenly exactly as it does in a real search.  Per scheduling round the script records the live trials of the
fullest rank -- a round lasts as long as its fullest GPU -- and prices the round with a per-step cost measured on one
MI355X as a function of the number of live trials (profiles/r1_v8_bench_per_epoch_summary.txt: 256 trials 433 us,
64 trials 185 us, 16 trials 111 us per step; linear in between).

    python bench/rebalance_sim.py [--ranks 8] [--trials-per-gpu 256] [--rounds 16]
"""
import argparse
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as headline  # noqa: E402
import mitune  # noqa: E402
from mitune.packed import PackedExecutor, PackedTrainable  # noqa: E402
from mitune.packed.engine import PackEngine  # noqa: E402
from mitune.parallel.comm import ThreadComm  # noqa: E402
from mitune.schedulers import ASHAScheduler  # noqa: E402

STEPS_PER_EPOCH = 3125                              # 800k training rows / batch 256
COST_US = [(0, 0.0), (16, 111.0), (64, 185.0), (256, 433.0)]


def step_cost_us(live: int) -> float:
    xs, ys = zip(*COST_US)
    return float(np.interp(live, xs, ys))


def true_cost(config) -> float:
    """Seconds a round of this trial adds to its GPU: HipPackEngine.trial_cost for the headline data (f32)."""
    w = [64, int(config["width_0"]), int(config["width_1"]), int(config["width_2"]), 1]
    macs = sum(a * b for a, b in zip(w[:-1], w[1:]))
    b = int(config["batch_size"])
    steps = 800_000 // b
    return macs * (6.0 * steps * b + 2.0 * 200_000) / 6.0e13 + steps * 2.0e-6


class ReplayEngine(PackEngine):
    name = "replay"

    def __init__(self, capacity: int, log: list, weigh: bool = True, loads: list = None):
        self.capacity, self.slots, self.log = capacity, {}, log
        self.weigh, self.loads, self.costs = weigh, loads if loads is not None else [], {}

    def trial_cost(self, config):
        return true_cost(config) if self.weigh else 1.0

    def load(self, slot, trial_index, config, state=None):
        self.slots[slot] = [int(trial_index), int((state or {}).get("epoch", 0))]
        self.costs[slot] = true_cost(config)

    def evict(self, slot):
        self.slots.pop(slot, None)

    def occupied(self):
        return sorted(self.slots)

    def train_epoch(self):
        self.log.append(len(self.slots))
        self.loads.append(sum(self.costs[k] for k in self.slots))
        for s in self.slots.values():
            s[1] += 1

    def evaluate(self):
        out = {}
        for slot, (idx, ep) in self.slots.items():
            quality = np.random.default_rng(idx).random()                     # a trial's talent ...
            noise = np.random.default_rng(idx * 1000 + ep).normal(0, 0.05)    # ... and this epoch's luck
            loss = 0.2 + quality / (1 + 0.3 * ep) + noise
            out[slot] = {"train_loss": loss, "validation_loss": loss, "validation_mse": loss, "validation_mape": 0.0}
        return out

    def export(self, slot):
        return {"model": {}, "epoch": self.slots[slot][1], "step": 0, "trial_index": self.slots[slot][0]}


def run(ranks: int, per_gpu: int, rounds: int, gap: int, weigh: bool = True, slack: int = 0):
    comms = ThreadComm.make(ranks)
    X = torch.zeros(8, 4)
    logs = [[] for _ in range(ranks)]
    loads = [[] for _ in range(ranks)]
    stats = [None] * ranks

    def body(r):
        cap = per_gpu + slack
        tr = PackedTrainable(train=(X, X[:, 0]), val=(X, X[:, 0]), max_epochs=rounds, capacity=cap, rebalance_gap=gap)
        ex = PackedExecutor(tr, comm=comms[r], device=torch.device("cpu"), capacity=cap, seed=0,
                            engine=ReplayEngine(cap, logs[r], weigh, loads[r]))
        sched = ASHAScheduler(metric="validation_loss", mode="min", max_t=rounds, grace_period=1, reduction_factor=4)
        an = mitune.run(tr, config=headline.search_space(), num_samples=per_gpu * ranks, scheduler=sched, executor=ex,
                        seed=0, verbose=0, local_dir=None)
        stats[r] = an.stats

    ths = [threading.Thread(target=body, args=(r,)) for r in range(ranks)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    live = np.array(logs)                                                     # [rank, round]
    run.loads = np.array(loads)                                               # modelled seconds per rank and round
    return live, stats[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--trials-per-gpu", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=16)
    args = ap.parse_args()
    one, _ = run(1, args.trials_per_gpu, args.rounds, 0)
    t_one = sum(step_cost_us(int(v)) for v in one[0]) * STEPS_PER_EPOCH / 1e6
    print(f"1 rank : live per round {one[0].tolist()}  modelled train time {t_one:.2f} s")
    for gap in (0, 2):
        live, st = run(args.ranks, args.trials_per_gpu, args.rounds, gap)
        worst, mean = live.max(0), live.mean(0)
        t = sum(step_cost_us(int(v)) for v in worst) * STEPS_PER_EPOCH / 1e6
        print(f"{args.ranks} ranks, rebalance_gap {gap}: fullest rank per round {worst.tolist()}")
        print(f"          mean per round {[round(float(m), 1) for m in mean]}  migrations {st['migrations']}  "
              f"modelled train time {t:.2f} s  -> weak-scaling efficiency of the training part {t_one / t:.3f}")
    # The same search priced by what each trial weighs (width, batch size) instead of by head count: a round lasts as long
    # as the rank whose trials weigh the most.
    for label, weigh, slack in (("trials dealt and moved by head count", False, 0), ("trials dealt and moved by weight", True, 0),
                                ("by weight, 8 spare slots per rank", True, 8)):
        live, st = run(args.ranks, args.trials_per_gpu, args.rounds, 2, weigh, slack)
        loads = run.loads
        worst, mean = loads.max(0), loads.mean(0)
        print(f"{args.ranks} ranks, {label}: heaviest / mean rank per round "
              f"{[round(float(a / b), 3) for a, b in zip(worst, mean)]}")
        print(f"          migrations {st['migrations']}  modelled round times: sum of heaviest {worst.sum():.2f} s, sum of mean "
              f"{mean.sum():.2f} s  -> weak-scaling efficiency of the training part {mean.sum() / worst.sum():.3f}")


if __name__ == "__main__":
    main()
