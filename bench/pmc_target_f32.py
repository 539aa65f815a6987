#!/usr/bin/env python
"""Fixed, short workload of the exact-f32 kernels for hardware-counter collection (run under ``rocprofv3 --pmc ...``).

The grouped f32 GEMM in its two regimes -- a full pack (256 slots, 512-wide layers, batch 512; 128-tiles) and a
late-round pack (16 slots; 64-tiles) -- in all three modes, then one epoch of a real 64-slot f32 MLP pack (head,
optimiser, gather kernels) and a few steps of a 16-slot sequence pack (grouped attention / LayerNorm kernels).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from gemm_micro import Problem        # noqa: E402
from mitune.ops import lib as L       # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    so = L.load()
    for slots, edge in ((256, 128), (16, 64)):
        for mode in (0, 1, 2):
            prob = Problem(mode, slots, 512, 512, edge=edge, dtype=torch.float32)
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(iters):
                so.mt_gemm_f32_launch(mode, prob.gt.data_ptr(), prob.tt.data_ptr(), prob.ntiles, 0, 1, 0, 0, 1,
                                      prob.slab if mode == 2 else 0, edge, stream)
            torch.cuda.synchronize()
    import bench as headline
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    from mitune.tune import space as sp
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(20480, 64, dev)
    tr = PackedTrainable(train=train, val=val, max_epochs=1, engine="hip", dtype="f32", capacity=64, val_chunk=4096,
                         model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
    eng = HipPackEngine(tr, dev, 64, 0)
    for i, cfg in enumerate(sp.generate_variants(headline.search_space(), 64, seed=0)):
        eng.load(i, i, cfg)
    eng.train_epoch()
    eng.evaluate()
    torch.cuda.synchronize()
    # a 16-slot pack of sample-space Transformer trials: 4 steps
    import random
    from mitune.packed.engine_seq import SeqPackEngine
    g = torch.Generator().manual_seed(0)
    X, y = torch.rand(168, 50, 10, generator=g), torch.rand(168, generator=g)
    st = PackedTrainable(model="simple_transformer", train=(X[:128], y[:128]), val=(X[128:], y[128:]), max_epochs=1,
                         dtype="f32", engine="seq", capacity=16, model_kwargs={"batch_size": 32, "optimizer": "adamw"})
    seq = SeqPackEngine(st, dev, 16, 0)
    rng = random.Random(0)
    for i in range(16):
        seq.load(i, i, {"d_model": rng.choice([32, 64, 128]), "num_heads": rng.choice([2, 4]),
                        "num_layers": rng.choice([1, 2, 3]), "dropout": 0.2, "learning_rate": 1e-3, "weight_decay": 1e-4})
    seq.train_epoch()
    seq.evaluate()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
