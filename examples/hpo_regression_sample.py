#!/usr/bin/env python
"""The sample HPO workflow: 10 random-search trials of a small Transformer regressor under ASHA.

Counterpart of /root/reference/ray-tune-hpo-regression-sample.py:152-172 with the imports switched from
``ray.tune`` to ``mitune.tune``; no cluster runtime, no hard-coded home directory.

    python examples/hpo_regression_sample.py [--num-samples 10] [--local-dir ./mitune_results] [--epochs 5] [--packed K]

``--packed K`` hands the same search to the packed executor: K trials resident on the GPU at a time, all of them advancing
in shared kernel launches (grouped GEMMs, grouped attention / LayerNorm; ``packed/engine_seq.py``) -- hundreds of times
the throughput of one trial at a time; the result tree on disk has the same layout, ``--checkpoint`` adds a
``checkpoint_<iter>/model.pt`` per trial that loads into the plain ``SimpleTransformerModel``.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mitune import tune                                     # noqa: E402
from mitune.data import create_dummy_data                   # noqa: E402
from mitune.schedulers import ASHAScheduler                 # noqa: E402
from mitune.train.functions import train_dummy_model        # noqa: E402
from mitune.utils.logging import setup_logging              # noqa: E402
from search_spaces import SAMPLE_SPACE                      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-samples", type=int, default=10)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--local-dir", default="./mitune_results")
    ap.add_argument("--noise-target", action="store_true", help="target independent of the inputs, like the reference")
    ap.add_argument("--packed", type=int, default=0, help="trial slots of the packed executor (0: function trainable)")
    ap.add_argument("--checkpoint", action="store_true", help="packed: write every trial's final weights")
    args = ap.parse_args()
    logger = setup_logging(args.local_dir, prefix="mitune_workflow_sample_run")
    logger.info("Starting the mitune workflow...")
    train_loader, val_loader = create_dummy_data(learnable=not args.noise_target)
    if args.packed:
        from mitune.packed import PackedTrainable
        trainable = PackedTrainable(model="simple_transformer", train=tuple(train_loader.dataset.tensors),
                                    val=tuple(val_loader.dataset.tensors), max_epochs=args.epochs, dtype="f32",
                                    capacity=args.packed, name="train_dummy_model_packed",
                                    checkpoint_at_end=args.checkpoint,
                                    model_kwargs={"batch_size": 32, "optimizer": "adamw"})
    else:
        trainable = tune.with_parameters(train_dummy_model, data_loader=train_loader, val_loader=val_loader,
                                         epochs=args.epochs)
    analysis = tune.run(
        trainable,
        config=SAMPLE_SPACE,
        num_samples=args.num_samples,
        scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=args.epochs),
        resources_per_trial={"cpu": 2, "gpu": 1},
        local_dir=args.local_dir,
        verbose=1,
    )
    best_config = analysis.best_config
    logger.info("Best hyperparameters found: \n %s", best_config)
    print("Best hyperparameters found: \n", best_config)
    print("Best validation_loss:", analysis.best_result["validation_loss"])


if __name__ == "__main__":
    main()
