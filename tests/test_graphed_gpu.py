"""GPU tests: small fused Transformer-path kernels, the device-state flat optimiser and graph-replayed training."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ------------------------------------------------------------------ positional encoding + dropout
def test_posenc_no_dropout_matches_reference():
    from mitune.models.transformer import sinusoid_table
    from mitune.ops.elementwise import pe_add_dropout, pe_add_dropout_reference
    x = _rand(3, 50, 64).requires_grad_()
    pe = sinusoid_table(96, 64).to(DEV)
    y = pe_add_dropout(x, pe, 0.3, training=False)
    assert torch.equal(y, pe_add_dropout_reference(x, pe))
    y.sum().backward()
    assert torch.equal(x.grad, torch.ones_like(x))


def test_posenc_dropout_mask_is_consistent_and_fresh():
    from mitune.models.transformer import sinusoid_table
    from mitune.ops.elementwise import pe_add_dropout
    p = 0.25
    x = (_rand(8, 50, 128).abs() + 1.0).requires_grad_()          # x + pe > 0 everywhere, so zeros are drops
    pe = sinusoid_table(50, 128).to(DEV)
    torch.manual_seed(0)
    y = pe_add_dropout(x, pe, p, training=True)
    kept = y != 0
    frac = kept.float().mean().item()
    assert abs(frac - (1 - p)) < 0.01
    ref = (x + pe) / (1 - p)
    assert torch.allclose(y[kept], ref[kept], rtol=1e-6, atol=1e-6)
    dy = _rand(8, 50, 128, seed=1)
    y.backward(dy)
    assert torch.allclose(x.grad, torch.where(kept, dy / (1 - p), torch.zeros_like(dy)), rtol=1e-6, atol=1e-7)
    y2 = pe_add_dropout(x, pe, p, training=True)
    assert not torch.equal(y2 != 0, kept)                            # a new seed, a new mask
    # drops are not clustered along rows or channels
    assert (kept.float().mean(dim=(0, 1)) - (1 - p)).abs().max() < 0.12
    assert (kept.float().mean(dim=2) - (1 - p)).abs().max() < 0.2


# ------------------------------------------------------------------ depthwise conv1d
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("B,S,d", [(3, 17, 64), (2, 96, 100), (32, 50, 128)])
def test_depthwise_conv1d_matches_reference(B, S, d, k):
    from mitune.ops.elementwise import depthwise_conv1d, depthwise_conv1d_reference
    x = _rand(B, S, d, seed=1).requires_grad_()
    w = _rand(d, 1, k, seed=2, scale=0.5).requires_grad_()
    b = _rand(d, seed=3).requires_grad_()
    dy = _rand(B, S, d, seed=4)
    y = depthwise_conv1d(x, w, b)
    y.backward(dy)
    got = (y.detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    x.grad = w.grad = b.grad = None
    yr = depthwise_conv1d_reference(x, w, b)
    yr.backward(dy)
    for a, r, name in zip(got, (yr.detach(), x.grad, w.grad, b.grad), ("y", "dx", "dw", "db")):
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-4 * max(1.0, r.abs().max().item())), name


# ------------------------------------------------------------------ fused head MLP
@pytest.mark.parametrize("B,d,dims", [(32, 64, (128, 64, 32, 16, 1)), (5, 512, (128, 64, 32, 16, 1)),
                                      (100, 192, (128, 64, 32, 16, 1)), (33, 40, (96, 24, 1)), (16, 128, (1,))])
def test_head_mlp_matches_reference(B, d, dims):
    from mitune.ops.head_mlp import head_mlp, head_mlp_reference
    seq = _rand(B, 7, d, seed=5).requires_grad_()
    ws, bs, k = [], [], d
    for i, o in enumerate(dims):
        ws.append(_rand(o, k, seed=10 + i, scale=1.0 / math.sqrt(k)).requires_grad_())
        bs.append(_rand(o, seed=20 + i, scale=0.1).requires_grad_())
        k = o
    dout = _rand(B, dims[-1], seed=6)
    out = head_mlp(seq[:, -1, :], ws, bs)                      # strided last-token view, read in place
    out.backward(dout)
    got = [out.detach(), seq.grad.clone()] + [w.grad.clone() for w in ws] + [b.grad.clone() for b in bs]
    seq.grad = None
    for t in ws + bs:
        t.grad = None
    ref = head_mlp_reference(seq[:, -1, :], ws, bs)
    ref.backward(dout)
    want = [ref.detach(), seq.grad] + [w.grad for w in ws] + [b.grad for b in bs]
    for i, (a, r) in enumerate(zip(got, want)):
        assert torch.allclose(a, r, rtol=2e-4, atol=2e-5 * max(1.0, r.abs().max().item())), i


# ------------------------------------------------------------------ flat optimiser vs torch.optim
@pytest.mark.parametrize("name", ["adam", "adamw", "sgd", "rmsprop"])
@pytest.mark.parametrize("clip", [0.0, 0.5])
def test_flat_optimizer_matches_torch(name, clip):
    from mitune.train.graphed import FlatOptimizer, FlatParams
    from mitune.train.optim import build_optimizer
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(13, 37), torch.nn.ReLU(), torch.nn.Linear(37, 3)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Linear(13, 37), torch.nn.ReLU(), torch.nn.Linear(37, 3)).to(DEV)
    b.load_state_dict(a.state_dict())
    flat = FlatParams(a)
    fo = FlatOptimizer(flat, name, lr=3e-3, weight_decay=1e-2, momentum=0.9, clip=clip)
    to = build_optimizer(name, b.parameters(), 3e-3, 1e-2, 0.9)
    for step in range(6):
        x, y = _rand(16, 13, seed=step), _rand(16, 3, seed=100 + step)
        F.mse_loss(a(x), y).backward()
        fo.step()
        to.zero_grad()
        F.mse_loss(b(x), y).backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(b.parameters(), clip)
        to.step()
    assert fo.steps_taken == 6
    assert float(flat.G.abs().max()) == 0.0                    # the update zeroed the gradient slab
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=2e-6)


def test_flat_optimizer_schedule_and_state_roundtrip():
    from mitune.models.mlp import lr_factor
    from mitune.train.graphed import FlatOptimizer, FlatParams
    m = torch.nn.Linear(4, 4, bias=False).to(DEV)
    flat = FlatParams(m)
    fo = FlatOptimizer(flat, "sgd", lr=0.1, momentum=0.0, warmup=3, total=10)
    w0 = m.weight.detach().clone()
    expect = w0.clone()
    for s in range(5):
        m.weight.grad.fill_(1.0)
        fo.step()
        expect -= 0.1 * lr_factor(s, 3, 10)
    assert torch.allclose(m.weight, expect, atol=1e-6)
    sd = fo.state_dict()
    fo2 = FlatOptimizer(FlatParams(torch.nn.Linear(4, 4, bias=False).to(DEV)), "sgd", lr=0.1, momentum=0.0)
    fo2.load_state_dict(sd)
    assert fo2.steps_taken == 5


# ------------------------------------------------------------------ graph-replayed training
def _toy(n=416, s=12, f=10, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n, s, f, generator=g)
    y = X[:, -1, 0] - X[:, 0, 1]
    return X.to(DEV), y.to(DEV)


def test_graphed_training_matches_eager_stepping():
    from mitune.models.transformer import SimpleTransformerModel
    from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
    X, y = _toy()
    losses = {}
    for use_graph in (True, False):
        torch.manual_seed(0)
        model = SimpleTransformerModel(10, 32, 2, 1, dropout=0.0).to(DEV)
        opt = FlatOptimizer(FlatParams(model), "adamw", lr=1e-3, weight_decay=1e-4)
        tr = GraphedTrainer(model, torch.nn.MSELoss(), X, y, 32, opt, seed=3, use_graph=use_graph)
        assert tr.graphed == use_graph
        losses[use_graph] = [tr.train_epoch() for _ in range(3)]
        assert opt.steps_taken == 3 * 13
    assert losses[True][-1] < losses[True][0]
    for a, b in zip(losses[True], losses[False]):
        assert abs(a - b) < 2e-3 * max(1.0, abs(b))


def test_graphed_training_with_dropout_and_ragged_tail_learns():
    from mitune.models.transformer import TransformerModel
    from mitune.train.graphed import FlatOptimizer, FlatParams, GraphedTrainer
    X, y = _toy(n=430)                                           # 13 full batches + a tail of 14 rows
    torch.manual_seed(1)
    model = TransformerModel(10, 2, 2, 32, 64, 0.1, attention_type="linear_attention",
                             depthwise_separable_conv=True).to(DEV)
    opt = FlatOptimizer(FlatParams(model), "adam", lr=2e-3, clip=1.0, warmup=5, total=400)
    tr = GraphedTrainer(model, torch.nn.MSELoss(), X, y.unsqueeze(1), 32, opt, seed=0)
    assert tr.graphed
    first = tr.train_epoch()
    for _ in range(5):
        last = tr.train_epoch()
    assert opt.steps_taken == 6 * 14 and math.isfinite(last) and last < first
    m = tr.evaluate(X, y.unsqueeze(1))
    assert math.isfinite(m["validation_loss"]) and math.isfinite(m["validation_mape"])


def test_sample_workflow_runs_graphed_on_gpu(tmp_path):
    from mitune import tune
    from mitune.data import create_dummy_data
    from mitune.schedulers import ASHAScheduler
    from mitune.train.functions import train_dummy_model
    tl, vl = create_dummy_data(num_samples=320, seq_len=20, learnable=True)
    an = tune.run(tune.with_parameters(train_dummy_model, data_loader=tl, val_loader=vl, epochs=3),
                  config={"d_model": tune.choice([32, 64]), "num_heads": 2, "num_layers": tune.choice([1, 2]),
                          "dropout": 0.1, "learning_rate": tune.loguniform(1e-3, 1e-2), "weight_decay": 1e-4},
                  num_samples=3, scheduler=ASHAScheduler(metric="validation_loss", mode="min"),
                  resources_per_trial={"cpu": 1, "gpu": 1}, local_dir=str(tmp_path), verbose=0)
    assert len(an.trials) == 3 and all(t.status == "TERMINATED" for t in an.trials)
    assert all(t.last_result["graphed"] for t in an.trials)
    assert an.best_config is not None
