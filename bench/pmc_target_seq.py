#!/usr/bin/env python
"""Fixed, short workload of the sequence-model kernels for hardware-counter collection (``rocprofv3 --pmc ...``):
a few eager training steps of the sample Transformer (f32: fused small-layer backward, f32-operand attention, narrow
LayerNorm, fused relu+dropout) and of an FT-Transformer block stack under bf16 autocast (bf16 attention, split-K weight
gradient)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mitune.data import create_dummy_data, tabular_regression          # noqa: E402
from mitune.models.ft_transformer import FTTransformer                 # noqa: E402
from mitune.models.transformer import SimpleTransformerModel           # noqa: E402
from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer   # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda")
    tl, _ = create_dummy_data()
    X, Y = (t.to(dev) for t in tl.dataset.tensors)
    for d, h, layers in ((64, 4, 2), (128, 4, 3)):
        model = SimpleTransformerModel(10, d, h, layers, 0.2).to(dev)
        opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adamw", 1e-3, 1e-4)
        tr = GraphedTrainer(model, torch.nn.MSELoss(), X, Y, 32, opt, use_graph=False)
        tr.begin_epoch()
        for i in range(steps):
            tr.step_once(i)
    (Xt, yt), _ = tabular_regression(8192, 64, seed=1, device=dev)
    model = FTTransformer(64, 192, 3, 8).to(dev)
    opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adamw", 1e-3, 1e-4)
    tr = GraphedTrainer(model, torch.nn.MSELoss(), Xt, yt, 1024, opt, use_graph=False,
                        forward_ctx=lambda: torch.autocast("cuda", dtype=torch.bfloat16))
    tr.begin_epoch()
    for i in range(steps):
        tr.step_once(i)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
