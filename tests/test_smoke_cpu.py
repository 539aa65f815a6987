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


def test_ft_transformer_population_based_training_cpu():
    from mitune.schedulers import PopulationBasedTraining
    g = torch.Generator().manual_seed(0)
    X = torch.randn(1200, 6, generator=g)
    y = X[:, 0] * X[:, 1] + torch.sin(X[:, 2]) + 0.05 * torch.randn(1200, generator=g)
    tr = PackedTrainable(model="ft_transformer", train=(X[:1000], y[:1000]), val=(X[1000:], y[1000:]), max_epochs=4,
                         engine="torch", capacity=4, model_kwargs={"d_token": 16, "n_blocks": 1, "n_heads": 2})
    pbt = PopulationBasedTraining(metric="validation_loss", mode="min", perturbation_interval=2,
                                  hyperparam_mutations={"learning_rate": tune.loguniform(1e-4, 1e-2)},
                                  quantile_fraction=0.25, seed=0)
    an = mitune.run(tr, config={"learning_rate": tune.loguniform(1e-4, 1e-2), "batch_size": 100,
                                "attention_dropout": 0.0}, num_samples=4, scheduler=pbt, verbose=0)
    assert all(t.training_iteration == 4 for t in an.trials)
    assert an.stats["exploits"] >= 1
    first = min(t.results[0]["validation_loss"] for t in an.trials)
    assert an.best_result["validation_loss"] < first


def test_ft_transformer_last_block_carries_only_the_summary_token():
    """The last block runs its output projection, LayerNorms and feed-forward on the summary token alone; prediction and
    every parameter gradient equal those of the block run over all tokens."""
    import torch
    from mitune.models.ft_transformer import FTTransformer
    torch.manual_seed(3)
    model = FTTransformer(7, d_token=16, n_blocks=2, n_heads=2, attn_dropout=0.0, ffn_dropout=0.0)
    x = torch.randn(5, 7)

    def full(m, x):
        t, _ = m.tokenizer(x)
        for blk in m.blocks:
            t, _ = blk(t, None, last=False)
        cls = t[:, -1, :]
        return torch.nn.functional.linear(torch.relu(m.head_norm(cls)), m.head.weight, m.head.bias).squeeze(-1)

    a = model(x)
    ga = torch.autograd.grad(a.square().sum(), list(model.parameters()))
    b = full(model, x)
    gb = torch.autograd.grad(b.square().sum(), list(model.parameters()))
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    for (name, _), u, v in zip(model.named_parameters(), ga, gb):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-5), name
