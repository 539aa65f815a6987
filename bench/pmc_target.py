#!/usr/bin/env python
"""Fixed, short kernel workload for hardware-counter collection (run under ``rocprofv3 --pmc ...``).

Launches each grouped-GEMM mode a few times on the two regimes the search spends its time in -- a full pack
(256 slots, 512-wide layers, batch 512; throughput form) and a late-ASHA pack (16 slots; latency form) -- plus
a handful of whole training steps of a real pack so the head / optimiser / gather kernels appear too.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from gemm_micro import Problem        # noqa: E402
from mitune.ops import lib as L       # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    so = L.load()
    for slots, nstage in ((256, 2), (16, 4)):
        for mode in (0, 1, 2):
            prob = Problem(mode, slots, 512, 512)
            stream = torch.cuda.current_stream().cuda_stream
            for _ in range(iters):
                so.mt_gemm_launch(mode, prob.gt.data_ptr(), prob.tt.data_ptr(), prob.ntiles, 0, 1, 0, 0, 1,
                                  prob.slab if mode == 2 else 0, nstage, stream)
            torch.cuda.synchronize()
    # a few real training steps (all kernels of the step driver)
    import bench as headline
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    from mitune.tune import space as sp
    dev = torch.device("cuda", 0)
    train, val = headline.make_data(40960, 64, dev)
    tr = PackedTrainable(train=train, val=val, max_epochs=1, engine="hip", capacity=64, val_chunk=4096,
                         model_kwargs={"max_width": 512, "max_layers": 3, "max_batch": 1024})
    eng = HipPackEngine(tr, dev, 64, 0)
    for i, cfg in enumerate(sp.generate_variants(headline.search_space(), 64, seed=0)):
        eng.load(i, i, cfg)
    eng.train_epoch()
    eng.evaluate()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
