#!/usr/bin/env python
"""The full HPO workflow: 50 trials of the configurable Transformer under ASHA + Bayesian optimisation.

Counterpart of /root/reference/ray-tune-hpo-regression.py:465-480.  The real wearable-sensor tables are
private; point ``--data-dir`` at ``<id>_combined_data.npy`` files to use them, otherwise a synthetic
stand-in with the same 81 feature columns is generated.

    python examples/hpo_regression.py --num-samples 50 --subject MMCS0002 [--data-dir DIR]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mitune import tune                                          # noqa: E402
from mitune.data import get_data_loaders                         # noqa: E402
from mitune.schedulers import ASHAScheduler                      # noqa: E402
from mitune.search import BayesOptSearch                         # noqa: E402
from mitune.train.functions import train_transformer_model       # noqa: E402
from search_spaces import FULL_SPACE                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subject", default="MMCS0002")
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--num-samples", type=int, default=50)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--local-dir", default="./mitune_results")
    ap.add_argument("--max-layers", type=int, default=None, help="cap num_encoder_layers (CPU smoke runs)")
    args = ap.parse_args()
    data_loader, val_loader = get_data_loaders(args.subject, data_dir=args.data_dir)
    space = dict(FULL_SPACE)
    if args.max_layers:
        space["num_encoder_layers"] = tune.choice([n for n in (2, 4, 6, 8, 12, 16) if n <= args.max_layers])
    analysis = tune.run(
        tune.with_parameters(train_transformer_model, data_loader=data_loader, val_loader=val_loader,
                             epochs=args.epochs),
        config=space,
        num_samples=args.num_samples,
        scheduler=ASHAScheduler(metric="validation_mape", mode="min"),
        search_alg=BayesOptSearch(metric="validation_mape", mode="min", random_search_steps=10),
        resources_per_trial={"cpu": 2, "gpu": 1},
        local_dir=args.local_dir,
        verbose=1,
    )
    print("Best hyperparameters found: \n", analysis.best_config)


if __name__ == "__main__":
    main()
