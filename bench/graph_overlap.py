#!/usr/bin/env python
"""Do captured training steps of different trials overlap when replayed on different HIP streams?

K sample-Transformer trainers (d_model 64, 4 heads, 2 layers, batch 32 x 50 tokens), 25 replays each: once slot after
slot on one stream, once round-robin over K streams issued by the same host thread.  Wall-clock, no profiler (kernel
tracing serialises dispatches).  usage: python bench/graph_overlap.py [K]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitune.data import create_dummy_data                                             # noqa: E402
from mitune.models.transformer import SimpleTransformerModel                          # noqa: E402
from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer            # noqa: E402


def main():
    dev = torch.device("cuda")
    tl, _ = create_dummy_data()
    X, Y = (t.to(dev) for t in tl.dataset.tensors)
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    trs = []
    for _ in range(K):
        model = SimpleTransformerModel(10, 64, 4, 2, 0.2).to(dev)
        opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adamw", 1e-3, 1e-4)
        trs.append(GraphedTrainer(model, torch.nn.MSELoss(), X, Y, 32, opt))
    for t in trs:
        t.train_epoch()
    torch.cuda.synchronize()
    pool = [torch.cuda.Stream() for _ in range(K + 8)]
    streams = pool[:K]

    def one_stream(n):
        t0 = time.perf_counter()
        for t in trs:
            t.pos.zero_()
            for _ in range(n):
                t.graph.replay()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, 0.0

    def many_streams(n):
        for t in trs:
            t.pos.zero_()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        t0 = time.perf_counter()
        for _ in range(n):
            for t, s in zip(trs, streams):
                with torch.cuda.stream(s):
                    t.graph.replay()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return time.perf_counter() - t0, host

    # which hardware queue a stream lands on is not visible: try every window of K consecutive pool streams
    for off in range(0, 9):
        streams[:] = pool[off:off + K]
        many_streams(25)
        a, _ = one_stream(25)
        b, host = many_streams(25)
        print(f"K={K} window {off}: one stream {a * 1e3:.1f} ms ({a / K / 25 * 1e6:.0f} us/step)  {K} streams {b * 1e3:.1f} ms "
              f"({b / K / 25 * 1e6:.0f} us/step)  speedup {a / b:.2f}  host issue {host * 1e3:.1f} ms", flush=True)


if __name__ == "__main__":
    main()
