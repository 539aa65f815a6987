"""GPU tests: small fused Transformer-path kernels, the device-state flat optimiser and graph-replayed training."""
import math
import os

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


# ------------------------------------------------------------------ flat optimiser vs torch.optim
@pytest.mark.parametrize("name", ["adam", "adamw", "sgd", "rmsprop"])
@pytest.mark.parametrize("clip", [0.0, 0.5])
@pytest.mark.parametrize("grad_slab", [True, False])
def test_flat_optimizer_matches_torch(name, clip, grad_slab):
    from mitune.train.graphed import FlatOptimizer, FlatParams
    from mitune.train.optim import build_optimizer
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(13, 37), torch.nn.ReLU(), torch.nn.Linear(37, 3)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Linear(13, 37), torch.nn.ReLU(), torch.nn.Linear(37, 3)).to(DEV)
    b.load_state_dict(a.state_dict())
    flat = FlatParams(a, grad_slab=grad_slab)
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
    if grad_slab:
        assert float(flat.G.abs().max()) == 0.0                # the update zeroed the gradient slab
    else:
        assert all(p.grad is None for p in a.parameters())     # gradients were read in place and released
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=2e-6)


def test_flat_optimizer_schedule_and_state_roundtrip():
    from mitune.models.mlp import lr_factor
    from mitune.train.graphed import FlatOptimizer, FlatParams
    m = torch.nn.Linear(4, 4, bias=False).to(DEV)
    flat = FlatParams(m)          # slab mode: the test writes gradients straight into the slab view
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
        opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adamw", lr=1e-3, weight_decay=1e-4)
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
    opt = FlatOptimizer(FlatParams(model, grad_slab=False), "adam", lr=2e-3, clip=1.0, warmup=5, total=400)
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


# ------------------------------------------------------------------ graph pack engine (FT-Transformer slots)
def _ft_trainable(engine):
    from mitune.data import tabular_regression
    from mitune.packed import PackedTrainable
    train, val = tabular_regression(4096, 16, seed=3, device=DEV)
    return PackedTrainable(model="ft_transformer", train=train, val=val, max_epochs=2, engine=engine, dtype="bf16",
                           capacity=4, model_kwargs={"d_token": 32, "n_blocks": 1, "n_heads": 4})


def test_graph_pack_engine_is_slot_invariant_and_matches_eager():
    from mitune.packed.engine import TorchPackEngine
    from mitune.packed.engine_graph import GraphPackEngine
    cfg = {"learning_rate": 2e-3, "weight_decay": 1e-4, "batch_size": 256, "optimizer": "adamw"}
    tr = _ft_trainable("graph")
    res = {}
    for slot in (0, 3):
        eng = GraphPackEngine(tr, DEV, 4, seed=0)
        eng.load(slot, 5, cfg)
        if slot == 3:
            eng.load(0, 9, dict(cfg, learning_rate=5e-3))           # a neighbour must not matter
        for _ in range(2):
            eng.train_epoch()
        res[slot] = eng.evaluate()[slot]
        assert eng._trainers[slot].graphed
    assert abs(res[0]["validation_mse"] - res[3]["validation_mse"]) <= 2e-3 * res[0]["validation_mse"]
    ref = TorchPackEngine(_ft_trainable("torch"), DEV, 4, seed=0)
    ref.load(0, 5, cfg)
    for _ in range(2):
        ref.train_epoch()
    want = ref.evaluate()[0]
    assert abs(res[0]["validation_mse"] - want["validation_mse"]) <= 0.1 * want["validation_mse"]      # bf16 autocast
    assert res[0]["validation_mse"] < 1.0


def test_graph_pack_engine_pbt_clone_and_search():
    import mitune
    from mitune import tune
    from mitune.packed.executor import PackedExecutor
    from mitune.schedulers import PopulationBasedTraining
    tr = _ft_trainable("auto")
    tr.max_epochs = 4
    ex = PackedExecutor(tr, device=torch.device(DEV), capacity=4, seed=0)
    assert ex.engine.name == "graph"
    pbt = PopulationBasedTraining(metric="validation_loss", mode="min", perturbation_interval=1,
                                  hyperparam_mutations={"learning_rate": tune.loguniform(1e-4, 1e-2)},
                                  quantile_fraction=0.25, seed=0)
    an = mitune.run(tr, config={"learning_rate": tune.loguniform(1e-4, 1e-2), "batch_size": 256, "optimizer": "adamw"},
                    num_samples=4, scheduler=pbt, executor=ex, seed=0, verbose=0, local_dir=None)
    assert len(an.trials) == 4 and all(t.training_iteration == 4 for t in an.trials)
    assert an.stats["exploits"] >= 1
    assert math.isfinite(an.best_result["validation_mse"])


# ------------------------------------------------------------------ counter-based dropout
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_counter_dropout_forward_backward_share_the_mask(dtype):
    from mitune.ops import rng
    from mitune.ops.elementwise import dropout
    p = 0.3
    x = (_rand(257, 131).abs() + 0.5).to(dtype).requires_grad_()       # odd sizes: exercises the ragged tail
    rng.seed_all(11)
    y = dropout(x, p, True)
    kept = y != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 0.01
    assert torch.allclose(y[kept].float(), (x.detach()[kept].float() / (1 - p)), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad != 0, kept)
    y2 = dropout(x, p, True)
    assert not torch.equal(y2 != 0, kept)                                # the counter moved
    rng.seed_all(11)
    assert torch.equal(dropout(x, p, True) != 0, kept)                   # and is reproducible when pinned
    assert dropout(x, p, False) is x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_relu_dropout_equals_dropout_of_relu(dtype):
    from mitune.ops import rng
    from mitune.ops.elementwise import dropout, relu_dropout
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(37, 50, 129, device=DEV, generator=g).to(dtype)
    dy = torch.randn(37, 50, 129, device=DEV, generator=g).to(dtype)
    a, b = x.clone().requires_grad_(), x.clone().requires_grad_()
    rng.seed_all(23)
    y = relu_dropout(a, 0.3, True)
    assert "FusedReluDropout" in type(y.grad_fn).__name__
    rng.seed_all(23)
    ref = dropout(torch.relu(b), 0.3, True)
    y.backward(dy)
    ref.backward(dy)
    assert torch.equal(y, ref) and torch.equal(a.grad, b.grad)
    assert relu_dropout(x, 0.3, False).equal(torch.relu(x))


@pytest.mark.parametrize("shape", [(50, 10), (7,), (96, 81)])
def test_cursor_batch_equals_index_select(shape):
    from mitune.ops.batch_hip import cursor_batch
    g = torch.Generator(device=DEV).manual_seed(1)
    N, B = 333, 32
    X = torch.randn((N,) + shape, device=DEV, generator=g)
    Y = torch.randn(N, device=DEV, generator=g)
    perm = torch.randperm(N, device=DEV, generator=g)
    for start in (0, 64, N - B):
        pos = torch.tensor([start], dtype=torch.int64, device=DEV)
        xb, yb = cursor_batch(perm, pos, X, Y, B)
        rows = perm[start:start + B]
        assert torch.equal(xb, X[rows]) and torch.equal(yb, Y[rows])


@pytest.mark.parametrize("n", [32, 1000, 5])
def test_fused_mse_matches_mse_loss(n):
    from mitune.ops.batch_hip import FusedMSE
    g = torch.Generator(device=DEV).manual_seed(n)
    out = torch.randn(n, device=DEV, generator=g, requires_grad=True)
    y = torch.randn(n, device=DEV, generator=g)
    running = torch.full((), 2.0, device=DEV)
    loss = FusedMSE.apply(out, y, running)
    (3.0 * loss).backward()
    ref_out = out.detach().clone().requires_grad_()
    ref = torch.nn.functional.mse_loss(ref_out, y)
    (3.0 * ref).backward()
    assert torch.allclose(loss, ref, rtol=1e-5) and torch.allclose(out.grad, ref_out.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(running, 2.0 + ref.detach(), rtol=1e-5)


def test_keep_mask_statistics():
    from mitune.ops.elementwise import keep_mask
    m = keep_mask((64, 8, 65, 65), 0.2, torch.device(DEV))
    assert m.dtype == torch.uint8 and set(m.unique().tolist()) <= {0, 1}
    f = m.float()
    assert abs(f.mean().item() - 0.8) < 0.005
    assert (f.mean(dim=(0, 1, 2)) - 0.8).abs().max() < 0.03              # no column structure
    assert (f.mean(dim=(1, 2, 3)) - 0.8).abs().max() < 0.03
    a, b = f[..., :-1].flatten(), f[..., 1:].flatten()                   # neighbours uncorrelated
    assert abs(((a - 0.8) * (b - 0.8)).mean().item()) < 0.002


def test_many_graphs_with_dropout_and_eager_work_between_epochs_stay_finite():
    """Three captured slots with dropout, eager validation between epochs: the scenario that produced NaNs with the
    framework generator's graph-registered state."""
    from mitune.data import tabular_regression
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_graph import GraphPackEngine
    train, val = tabular_regression(40960, 64, seed=5, device=DEV)
    tr = PackedTrainable(model="ft_transformer", train=train, val=val, max_epochs=2, engine="graph", dtype="bf16", capacity=3,
                         model_kwargs={"d_token": 192, "n_blocks": 3, "n_heads": 8})
    eng = GraphPackEngine(tr, DEV, 3, seed=0)
    for i in range(3):
        eng.load(i, i, {"learning_rate": 3e-4, "weight_decay": 1e-5, "batch_size": 1024, "optimizer": "adamw"})
    for _ in range(2):
        eng.train_epoch()
        ev = eng.evaluate()
        assert all(math.isfinite(v["validation_mse"]) and math.isfinite(v["train_loss"]) for v in ev.values())


def _sequence_pack(streams, n_slots=5, epochs=2):
    from mitune.data import create_dummy_data
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_graph import GraphPackEngine
    tl, vl = create_dummy_data(num_samples=400, seq_len=20, input_size=6)
    tr = PackedTrainable(model="simple_transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                         max_epochs=epochs, engine="graph", dtype="f32", capacity=n_slots,
                         model_kwargs={"batch_size": 32, "optimizer": "adamw", "streams": streams})
    eng = GraphPackEngine(tr, DEV, n_slots, seed=3)
    shapes = [(32, 2, 1), (64, 4, 2), (32, 4, 3), (64, 2, 1), (128, 4, 2)]
    for i in range(n_slots):
        d, h, L = shapes[i % len(shapes)]
        eng.load(i, i, {"d_model": d, "num_heads": h, "num_layers": L, "dropout": 0.2, "learning_rate": 1e-3,
                        "weight_decay": 1e-4})
    out = []
    for _ in range(epochs):
        eng.train_epoch()
        out.append(eng.evaluate())
    return eng, out


def test_sequence_trials_issued_over_streams_equal_issue_on_one_stream():
    """Slots of different shapes replayed round-robin over four streams end where the same slots end when they are
    replayed one after the other: own dropout counter, cursor and loss accumulator per slot."""
    eng4, multi = _sequence_pack(4)
    assert len(eng4._issue_streams(eng4.occupied(), 32)) == 4 and eng4._calibrated
    eng1, single = _sequence_pack(1)
    assert len(eng1._issue_streams(eng1.occupied(), 32)) == 1
    for a, b in zip(multi, single):
        assert a.keys() == b.keys()
        for k in a:
            assert math.isfinite(a[k]["validation_loss"])
            assert a[k]["validation_loss"] == pytest.approx(b[k]["validation_loss"], rel=2e-3)
            assert a[k]["train_loss"] == pytest.approx(b[k]["train_loss"], rel=2e-3)
    assert multi[-1][0]["validation_loss"] < multi[0][0]["validation_loss"] * 1.5
    _, again = _sequence_pack(4)                          # concurrent replays share nothing: bitwise repeatable
    assert again == multi


def test_long_token_packs_stay_on_one_stream_unless_streams_are_asked_for():
    from mitune.data import tabular_regression
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_graph import GraphPackEngine
    train, val = tabular_regression(4096, 64, seed=5, device=DEV)
    tr = PackedTrainable(model="ft_transformer", train=train, val=val, max_epochs=1, engine="graph", dtype="bf16", capacity=2,
                         model_kwargs={"d_token": 64, "n_blocks": 1, "n_heads": 4})
    eng = GraphPackEngine(tr, DEV, 2, seed=0)
    cur = torch.cuda.current_stream()
    from mitune.packed.engine_graph import DEFAULT_STREAMS
    assert eng._issue_streams([0, 1], 1024) == [cur]                   # 1024 x 65 token rows per step: nothing to overlap
    assert len(eng._issue_streams([0, 1], 32)) == DEFAULT_STREAMS      # small steps overlap, bf16 or not
    eng._streams_forced = True                                         # model_kwargs["streams"] given: honoured for any pack
    assert len(eng._issue_streams([0, 1], 1024)) == DEFAULT_STREAMS


def test_packed_search_over_the_sample_space_on_gpu():
    from mitune import tune
    from mitune.data import create_dummy_data
    from mitune.packed import PackedTrainable
    from mitune.schedulers import ASHAScheduler
    tl, vl = create_dummy_data(num_samples=400, seq_len=20, input_size=6)
    tr = PackedTrainable(model="simple_transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                         max_epochs=3, dtype="f32", engine="graph", capacity=4, model_kwargs={"batch_size": 32, "optimizer": "adamw"})
    an = tune.run(tr, config={"d_model": tune.choice([32, 64]), "num_heads": tune.choice([2, 4]),
                              "num_layers": tune.choice([1, 2]), "dropout": tune.uniform(0.1, 0.3),
                              "learning_rate": tune.loguniform(1e-4, 1e-2), "weight_decay": 1e-4},
                  num_samples=6, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=3),
                  local_dir=None, verbose=0, seed=1, metric="validation_loss", mode="min", device=torch.device(DEV))
    assert len(an.trials) == 6 and an.stats["engine"] == "graph"
    assert math.isfinite(an.best_result["validation_loss"])


def test_trials_prepared_during_a_running_epoch_end_like_trials_built_on_demand(monkeypatch):
    from mitune import tune
    from mitune.data import create_dummy_data
    from mitune.packed import PackedTrainable
    from mitune.search.searcher import BasicVariantGenerator
    tl, vl = create_dummy_data(num_samples=400, seq_len=20, input_size=6)

    def search():
        tr = PackedTrainable(model="simple_transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                             max_epochs=2, dtype="f32", engine="graph", capacity=3, model_kwargs={"batch_size": 32, "optimizer": "adamw"})
        an = tune.run(tr, config={"d_model": tune.choice([32, 64]), "num_heads": 2, "num_layers": tune.choice([1, 2]),
                                  "dropout": 0.1, "learning_rate": tune.loguniform(1e-4, 1e-2)},
                      num_samples=9, local_dir=None, verbose=0, seed=2, metric="validation_loss", mode="min",
                      device=torch.device(DEV))
        return an, {t.trial_id: (t.config["d_model"], t.last_result["validation_loss"]) for t in an.trials}

    an_a, ahead = search()
    assert an_a.stats["prepared_ahead"] > 0
    monkeypatch.setattr(BasicVariantGenerator, "lookahead_safe", False)
    an_b, on_demand = search()
    assert an_b.stats["prepared_ahead"] == 0
    assert ahead.keys() == on_demand.keys()
    for k in ahead:
        assert ahead[k][0] == on_demand[k][0]
        assert ahead[k][1] == pytest.approx(on_demand[k][1], rel=2e-3)


def test_bf16_shadow_of_the_parameter_slab_tracks_updates_and_feeds_the_dense_layers():
    from mitune.ops.linear import linear
    from mitune.train.graphed import FlatOptimizer, FlatParams
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 48).to(DEV)
    flat = FlatParams(lin, grad_slab=False, bf16_shadow=True)
    opt = FlatOptimizer(flat, "adamw", 1e-2, 0.0)
    x = torch.randn(8192, 64, device=DEV)
    dy = torch.randn(8192, 48, device=DEV)
    assert lin.weight._mt_bf16.dtype == torch.bfloat16 and torch.equal(lin.weight._mt_bf16, lin.weight.detach().bfloat16())
    for _ in range(2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = linear(x, lin.weight, lin.bias)
            assert "DenseBF16" in type(y.grad_fn).__name__ and y.dtype == torch.bfloat16
        # products of the bf16-rounded x and the shadow (exactly the cast of the master), f32 accumulation, f32 bias
        ref = x.bfloat16().float() @ lin.weight.detach().bfloat16().float().t() + lin.bias.detach()
        assert torch.allclose(y.detach().float(), ref, rtol=1e-2, atol=1e-2)
        y.backward(dy.to(y.dtype))
        assert lin.weight.grad.dtype == torch.float32
        before = lin.weight.detach().clone()
        opt.step()
        assert not torch.equal(lin.weight.detach(), before)
        assert torch.equal(lin.weight._mt_bf16, lin.weight.detach().bfloat16())
        assert torch.equal(lin.bias._mt_bf16, lin.bias.detach().bfloat16())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        z = linear(x, lin.weight, lin.bias)
    ref = x.bfloat16().float() @ lin.weight.detach().bfloat16().float().t() + lin.bias.detach()
    assert z.dtype == torch.bfloat16 and torch.allclose(z.float(), ref, rtol=1e-2, atol=1e-2)


# ------------------------------------------------------------------ split-contraction weight gradient
@pytest.mark.parametrize("M,in_f,out_f", [(66560, 192, 576), (8192, 256, 192), (5000, 72, 40)])
def test_split_k_linear_matches_reference(M, in_f, out_f):
    from mitune.ops.linear import linear
    x = _rand(M, in_f, seed=1).requires_grad_()
    w = _rand(out_f, in_f, seed=2, scale=in_f ** -0.5).requires_grad_()
    b = _rand(out_f, seed=3, scale=0.1).requires_grad_()
    dy = _rand(M, out_f, seed=4)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = linear(x, w, b)
        assert "DenseBF16" in type(y.grad_fn).__name__          # forward and input gradient on own tiles too
    y.backward(dy.to(y.dtype))
    got = (y.float().detach(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    assert got[2].dtype == torch.float32
    x.grad = w.grad = b.grad = None
    # reference on the same bf16-rounded operands, accumulated in f32
    xb, wb, dyb = x.detach().bfloat16().float(), w.detach().bfloat16().float(), dy.bfloat16().float()
    ref_y = xb @ wb.t() + b.detach().bfloat16().float()
    ref_dw, ref_db, ref_dx = dyb.t() @ xb, dyb.sum(0), dyb @ wb
    assert torch.allclose(got[0], ref_y, rtol=2e-2, atol=2e-2 * ref_y.abs().max().item())
    assert torch.allclose(got[2], ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    assert torch.allclose(got[3], ref_db, rtol=1e-3, atol=1e-3 * ref_db.abs().max().item())
    assert torch.allclose(got[1], ref_dx, rtol=2e-2, atol=2e-2 * ref_dx.abs().max().item())


# ------------------------------------------------------------------ one process per trial on the GPU
def _gpu_trial(config, data=None):
    import torch as _t
    from mitune import tune as _tune
    X, y = data
    dev = _t.device("cuda:0")             # the trial's own GPU is always device 0 of its process
    w = _t.zeros(X.shape[1], device=dev, requires_grad=True)
    opt = _t.optim.SGD([w], lr=config["lr"])
    Xd, yd = X.to(dev), y.to(dev)
    for _ in range(3):
        loss = ((Xd @ w - yd) ** 2).mean()
        opt.zero_grad(); loss.backward(); opt.step()
        _tune.report(loss=float(loss), visible=os.environ.get("HIP_VISIBLE_DEVICES", ""))


def test_process_executor_pins_a_gpu_per_trial(tmp_path):
    from mitune import tune
    g = torch.Generator().manual_seed(0)
    X = torch.randn(256, 4, generator=g)
    y = X @ torch.tensor([1.0, -2.0, 0.5, 0.0])
    an = tune.run(tune.with_parameters(_gpu_trial, data=(X, y)), config={"lr": tune.grid_search([0.05, 0.1])},
                  resources_per_trial={"cpu": 1, "gpu": 1}, executor="process", local_dir=str(tmp_path), verbose=0,
                  metric="loss", mode="min")
    assert [t.status for t in an.trials] == ["TERMINATED", "TERMINATED"]
    assert all(t.training_iteration == 3 and t.last_result["visible"] == "0" for t in an.trials)
    assert an.best_result["loss"] < an.trials[0].results[0]["loss"]


# ------------------------------------------------------------------ two ranks, replicated scheduler, graph engine
def _two_rank_worker(rank, world, port, out_dir):
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import mitune
    from mitune import tune
    from mitune.data import create_dummy_data
    from mitune.packed import PackedExecutor, PackedTrainable
    from mitune.parallel.comm import TorchDistComm
    from mitune.schedulers import ASHAScheduler
    dev = torch.device("cuda", 0)                  # both ranks share the one GPU of the test box
    torch.cuda.set_device(dev)
    comm = TorchDistComm(backend="gloo", device=torch.device("cpu"), rank=rank, world_size=world)
    tl, vl = create_dummy_data(num_samples=300, seq_len=16, input_size=6)
    tr = PackedTrainable(model="simple_transformer", train=tuple(tl.dataset.tensors), val=tuple(vl.dataset.tensors),
                         max_epochs=3, dtype="f32", engine="graph", capacity=3, model_kwargs={"batch_size": 32, "optimizer": "adamw"})
    ex = PackedExecutor(tr, comm=comm, device=dev, capacity=3, seed=0)
    an = mitune.run(tr, config={"d_model": tune.choice([32, 64]), "num_heads": 2, "num_layers": tune.choice([1, 2]),
                                "dropout": 0.1, "learning_rate": tune.loguniform(1e-4, 1e-2)},
                    num_samples=8, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=3, reduction_factor=2),
                    executor=ex, seed=0, verbose=0, metric="validation_loss", mode="min")
    res = {t.trial_id: [t.training_iteration, t.last_result["validation_loss"], t.placement["rank"]] for t in an.trials}
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump({"res": res, "engine": an.stats["engine"]}, f)
    comm.close()


def test_two_ranks_with_the_graph_engine_reach_the_same_decisions(tmp_path):
    """Replicated searcher and scheduler over a gloo metric exchange, each rank training its own slots with captured
    steps on the GPU: both ranks end with the same table, and both owned trials."""
    import json
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = json.load(open(tmp_path / "rank0.json"))
    b = json.load(open(tmp_path / "rank1.json"))
    assert a["engine"] == b["engine"] == "graph"
    assert a["res"] == b["res"] and len(a["res"]) == 8
    assert {v[2] for v in a["res"].values()} == {0, 1}
    assert all(math.isfinite(v[1]) for v in a["res"].values())


@pytest.mark.timeout(150)
def test_pack_of_slow_steps_over_several_streams_completes():
    """Three bf16 FT-Transformer slots of 66 560 token rows per step, 78 steps each, replayed over three streams.  With
    the host free to enqueue all of it at once this pack used to stall for good after a dozen finished steps (a graph
    launch blocked in the runtime, no stream progressing); the engine now keeps a bounded number of replays in flight
    per stream.  One epoch must finish, and multi-stream issue must not change any slot's numbers."""
    from mitune.data import tabular_regression
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_graph import GraphPackEngine
    dev = torch.device("cuda", 0)
    train, val = tabular_regression(100_000, 64, seed=5, device=dev)
    cfg = {"learning_rate": 1e-3, "weight_decay": 1e-5, "batch_size": 1024, "optimizer": "adamw"}
    out = {}
    for streams in (3, 1):
        tr = PackedTrainable(model="ft_transformer", train=train, val=val, max_epochs=1, engine="graph", dtype="bf16",
                             capacity=3, model_kwargs={"d_token": 192, "n_blocks": 3, "n_heads": 8, "streams": streams})
        eng = GraphPackEngine(tr, "cuda", 3, 0)
        for s in range(3):
            eng.load(s, s, cfg)
        eng.train_epoch()
        out[streams] = eng.evaluate()
        if streams == 3:
            assert len(eng._streams) == 3
        del eng
    for s in range(3):
        assert out[3][s]["train_loss"] == out[1][s]["train_loss"] and out[3][s]["validation_loss"] == out[1][s]["validation_loss"]
        assert out[3][s]["validation_loss"] == out[3][s]["validation_loss"]       # finite
