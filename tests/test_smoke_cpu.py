import numpy as np
import torch

import mitune
from mitune import tune
from mitune.schedulers import ASHAScheduler
from mitune.packed import PackedTrainable


def test_function_trainable_random_search(tmp_path):
    def trainable(config, offset):
        for i in range(5):
            tune.report(validation_loss=(config["x"] - 0.3) ** 2 + offset + 1.0 / (i + 1))

    an = mitune.run(mitune.with_parameters(trainable, offset=0.5),
                    config={"x": tune.uniform(0, 1), "k": tune.choice([1, 2])},
                    num_samples=8, scheduler=ASHAScheduler(metric="validation_loss", mode="min"),
                    local_dir=str(tmp_path), verbose=0)
    assert len(an.trials) == 8
    best = an.best_config
    assert abs(best["x"] - 0.3) == min(abs(t.config["x"] - 0.3) for t in an.trials if t.training_iteration == 5)


def test_packed_mlp_cpu(tmp_path):
    g = torch.Generator().manual_seed(0)
    X = torch.randn(2000, 8, generator=g)
    y = (X[:, 0] * 2 - X[:, 1] + 0.5 * X[:, 2] * X[:, 3]) + 0.05 * torch.randn(2000, generator=g)
    tr = PackedTrainable(train=(X[:1600], y[:1600]), val=(X[1600:], y[1600:]), max_epochs=4, engine="torch",
                         capacity=4, checkpoint_at_end=True)
    an = mitune.run(tr, config={"width": tune.choice([16, 32]), "num_layers": 2,
                                "learning_rate": tune.loguniform(1e-3, 1e-2), "batch_size": 64},
                    num_samples=6, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=4,
                                                           reduction_factor=2),
                    local_dir=str(tmp_path), verbose=0)
    assert len(an.trials) == 6
    assert an.best_result["validation_loss"] < float(y.var())
    assert an.best_checkpoint is not None
