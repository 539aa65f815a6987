"""The program engine (grouped launches for the configurable Transformer family) against independent trainings.

Oracle: ``TorchPackEngine`` on the CPU -- one ordinary ``TransformerModel`` + ``torch.optim`` optimiser per trial, plain
f32 PyTorch ops -- fed the same initial weights and the same keyed-permutation batches.  Every product of the engine is
f32 (exact-f32 MFMA in the dense layers, f32 multiply-add chains or exact-f32 MFMA in attention), so with dropout and
stochastic depth off a short training agrees to f32 rounding; with them on the masks are the engine's own counter hash, and
what is checked is that a trial's numbers depend on neither its slot nor its neighbours (bitwise), that evaluation is
deterministic, and that training still converges.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, F = 24, 10
BASE = {"dropout": 0.0, "stochastic_depth_rate": 0.0, "learning_rate": 1e-3, "weight_decay": 1e-4, "batch_size": 16,
        "warmup_steps": 0, "total_steps": 0, "gradient_clipping": 0.0, "loss_function": "mse", "optimizer": "adamw",
        "max_seq_length": 50, "attn_kernel_size": 3, "key_dim_scaling": 1.0, "depthwise_separable_conv": False,
        "shared_weights": False, "attention_type": "scaled_dot_product"}
CONFIGS = [
    dict(BASE, d_model=64, num_heads=2, num_encoder_layers=2, dim_feedforward=128),
    dict(BASE, d_model=64, num_heads=4, num_encoder_layers=3, dim_feedforward=192, attention_type="linear_attention",
         attn_kernel_size=5, key_dim_scaling=0.5, optimizer="adam", loss_function="huber"),
    dict(BASE, d_model=128, num_heads=8, num_encoder_layers=2, dim_feedforward=256, depthwise_separable_conv=True,
         optimizer="rmsprop", learning_rate=3e-4, loss_function="mae", batch_size=32),
    dict(BASE, d_model=64, num_heads=32, num_encoder_layers=4, dim_feedforward=128, shared_weights=True, key_dim_scaling=0.25,
         optimizer="sgd", learning_rate=1e-2, gradient_clipping=0.5, warmup_steps=4, total_steps=100),
    # heads wider than the fused attention kernels take: explicit score matrices
    dict(BASE, d_model=192, num_heads=2, num_encoder_layers=2, dim_feedforward=384, key_dim_scaling=0.5, loss_function="mape"),
    dict(BASE, d_model=192, num_heads=2, num_encoder_layers=2, dim_feedforward=384, attention_type="linear_attention",
         attn_kernel_size=7, shared_weights=True, depthwise_separable_conv=True, batch_size=32),
]


def _data(n_train=64, n_val=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n_train + n_val, S, F, generator=g)
    w = torch.randn(F, generator=g) / F ** 0.5
    y = 2.0 + torch.tanh((X[:, -5:, :] @ w).mean(1)) + 0.05 * torch.randn(n_train + n_val, generator=g)
    return (X[:n_train], y[:n_train]), (X[n_train:], y[n_train:])


def _trainable(engine="xf", **kw):
    from mitune.packed import PackedTrainable
    train, val = _data()
    mk = {"max_d_model": 192, "max_layers": 4, "max_dim_feedforward": 384, "max_batch": 32}
    mk.update(kw.pop("model_kwargs", {}))
    return PackedTrainable(model="transformer", train=train, val=val, max_epochs=4, dtype="f32", engine=engine,
                           capacity=kw.pop("capacity", 8), val_chunk=16, model_kwargs=mk, **kw)


def _xf(tr, seed=3):
    from mitune.packed.engine_xf import XfPackEngine
    return XfPackEngine(tr, "cuda", tr.capacity, seed)


def test_layout_covers_every_parameter_of_the_module():
    from mitune.models.transformer import TransformerModel
    from mitune.packed.engine_xf import XfLayout, xf_spec
    tr = _trainable()
    for cfg in CONFIGS:
        model = TransformerModel.from_config(cfg, F)
        lay = XfLayout(F, 12, xf_spec(tr, cfg, 64))
        sd = model.state_dict()
        assert {e[0] for e in lay.entries} == set(sd.keys())
        for key, off, shape, rows, cols, ld in lay.entries:
            assert tuple(sd[key].shape) == tuple(shape) and sd[key].numel() == rows * cols and off % 8 == 0 and ld >= cols
        ends = sorted((off, off + rows * ld) for _, off, _, rows, _, ld in lay.entries)
        assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= lay.total


@pytest.mark.parametrize("which", list(range(len(CONFIGS))))
def test_one_trial_matches_its_independent_eager_training(which):
    """Dropout and stochastic depth off: two epochs of the engine against autograd + torch.optim on the CPU."""
    from mitune.packed.engine import TorchPackEngine
    tr = _trainable()
    eng = _xf(tr)
    ref = TorchPackEngine(copy.copy(tr), "cpu", tr.capacity, 3)
    cfg = CONFIGS[which]
    eng.load(2, 20 + which, cfg)
    ref.load(2, 20 + which, cfg)
    for k, v in ref.export(2)["model"].items():
        assert torch.equal(v, eng.export(2)["model"][k]), k            # same start
    for _ in range(2):
        eng.train_epoch()
        ref.train_epoch()
    got, want = eng.evaluate()[2], ref.evaluate()[2]
    for key in ("train_loss", "validation_mse", "validation_loss", "validation_mape"):
        assert got[key] == pytest.approx(want[key], rel=2e-3, abs=2e-5), (key, got, want)
    a, b = eng.export(2)["model"], ref.export(2)["model"]
    for k in b:
        if k.endswith("qkv_proj.bias"):
            continue          # the key bias has a mathematically zero gradient under softmax: Adam amplifies its rounding noise
        # Adam-family steps are lr * sign-like for tiny gradients (query / key weights, input columns that barely matter),
        # so rounding noise in such a gradient moves a weight by up to lr per step: the steps taken bound what may differ
        # there.  The gradients themselves are held to float64 autograd in the next test.
        steps = 2 * (64 // cfg["batch_size"])
        atol = 2e-3 if cfg["optimizer"] == "sgd" else max(2e-3, 0.9 * cfg["learning_rate"] * steps)
        assert torch.allclose(a[k], b[k], atol=atol, rtol=2e-2), (k, (a[k] - b[k]).abs().max())


@pytest.mark.parametrize("which", list(range(len(CONFIGS))))
def test_one_sgd_step_gives_the_float64_gradient_of_every_parameter(which):
    """lr = 1, no momentum, decay, clipping or schedule: parameters before minus after one step are the gradients the engine
    computed; the oracle is autograd on the plain module in float64 (dropout and stochastic depth off)."""
    from mitune.models.transformer import TransformerModel
    from mitune.packed.permutation import permute, round_keys
    from mitune.train.losses import get_criterion
    tr = _trainable()
    eng = _xf(tr)
    cfg = dict(CONFIGS[which], optimizer="sgd", momentum=0.0, learning_rate=1.0, weight_decay=0.0, gradient_clipping=0.0,
               warmup_steps=0, total_steps=0, batch_size=64)
    tr.model_kwargs["max_batch"] = 64
    eng = _xf(tr)
    eng.load(1, 10 + which, cfg)
    before = eng.export(1)["model"]
    eng.train_epoch()
    torch.cuda.synchronize()
    after = eng.export(1)["model"]
    X, y = tr.train
    model = TransformerModel.from_config(cfg, F).double()
    model.load_state_dict({k: v.double() for k, v in before.items()})
    model.train()
    idx = torch.from_numpy(permute(np.arange(64, dtype=np.uint32), 64, round_keys(3, 10 + which, 0)).astype(np.int64))
    loss = get_criterion(cfg["loss_function"])(model(X[idx].double()).reshape(-1), y[idx].double())
    loss.backward()
    for name, p in model.named_parameters():
        got = (before[name] - after[name]).double()
        ref = p.grad
        scale = ref.abs().max().item() + 1e-9
        err = (got - ref).abs().max().item()
        assert err <= 2e-4 * scale + 1e-7, f"{name}: |got - ref| max {err:.3e}, |ref| max {scale:.3e}"


def test_a_mixed_pack_matches_the_trials_trained_alone_bitwise():
    """All six shapes resident together, dropout and stochastic depth ON: every trial's weights and metrics are those of
    the same trial trained alone in another slot -- grouped launches, shared tables and neighbours change nothing."""
    tr = _trainable()
    cfgs = [dict(c, dropout=0.2, stochastic_depth_rate=0.15) for c in CONFIGS]
    pack = _xf(tr)
    for slot, (i, c) in zip((5, 0, 3, 2, 7, 1), enumerate(cfgs)):
        pack.load(slot, 30 + i, c)
    for _ in range(2):
        pack.train_epoch()
    m_pack = pack.evaluate()
    for slot, (i, c) in zip((5, 0, 3, 2, 7, 1), enumerate(cfgs)):
        alone = _xf(tr)
        alone.load(4, 30 + i, c)
        for _ in range(2):
            alone.train_epoch()
        m = alone.evaluate()[4]
        assert m == m_pack[slot], (i, m, m_pack[slot])
        wa, wp = alone.export(4)["model"], pack.export(slot)["model"]
        for k in wa:
            assert torch.equal(wa[k], wp[k]), (i, k)
        assert np.isfinite(m["train_loss"]) and np.isfinite(m["validation_mape"])


def test_a_pack_trains_to_the_same_bits_on_32_tiles_and_on_64_tiles(monkeypatch):
    """The tile edge of a grouped GEMM launch follows the launch's size, i.e. who else is resident; a trial's weights and
    metrics must not (same contraction order on every edge, weight-gradient splits chosen from the layer's own shape)."""
    from mitune.options import OPTIONS
    cfgs = [dict(c, dropout=0.1) for c in CONFIGS]
    out = []
    for tiny in (0, 10 ** 9):                  # never 32-tiles / 32-tiles for every launch below the 128-tile threshold
        monkeypatch.setattr(OPTIONS, "xf_tiny_tiles", tiny)
        pack = _xf(_trainable())
        for slot, (i, c) in zip((5, 0, 3, 2, 7, 1), enumerate(cfgs)):
            pack.load(slot, 30 + i, c)
        for _ in range(2):
            pack.train_epoch()
        out.append((pack.evaluate(), {slot: pack.export(slot)["model"] for slot in (5, 0, 3, 2, 7, 1)}))
    (m64, w64), (m32, w32) = out
    assert m64 == m32
    for slot in w64:
        for k in w64[slot]:
            assert torch.equal(w64[slot][k], w32[slot][k]), (slot, k)


def test_dropout_and_stochastic_depth_change_training_and_not_evaluation():
    tr = _trainable()
    eng = _xf(tr)
    eng.load(0, 40, dict(CONFIGS[0], dropout=0.3))
    eng.load(1, 40, dict(CONFIGS[0], dropout=0.0))
    eng.load(2, 40, dict(CONFIGS[0], num_encoder_layers=4, stochastic_depth_rate=0.5))
    eng.load(3, 40, dict(CONFIGS[0], num_encoder_layers=4, stochastic_depth_rate=0.0))
    before = [eng.evaluate()[k]["validation_mse"] for k in range(4)]
    assert before[0] == before[1] and before[2] == before[3]            # evaluation draws no mask
    eng.train_epoch()
    a, b = eng.evaluate(), eng.evaluate()
    assert a == b
    assert a[0]["train_loss"] != a[1]["train_loss"] and a[2]["train_loss"] != a[3]["train_loss"]
    assert all(np.isfinite(a[k]["validation_mse"]) for k in range(4))


def test_export_load_round_trip_and_clone():
    from mitune.models.transformer import TransformerModel
    tr = _trainable()
    cfg = dict(CONFIGS[1], dropout=0.1)
    a = _xf(tr)
    a.load(0, 50, cfg)
    a.train_epoch()
    state = a.export(0)
    a.train_epoch()
    want = a.evaluate()[0]
    b = _xf(tr)
    b.load(4, 50, cfg, state=state)
    b.train_epoch()
    assert b.evaluate()[4] == want
    TransformerModel.from_config(cfg, F).load_state_dict(state["model"])       # a plain module takes the exported weights
    a.load(1, 51, cfg)
    a.clone(0, 1, dict(cfg, learning_rate=1e-4))
    x, y = a.export(0), a.export(1)
    for k in x["model"]:
        assert torch.equal(x["model"][k], y["model"][k]) and torch.equal(x["optimizer"]["exp_avg"][k], y["optimizer"]["exp_avg"][k])
    assert a.slots[1].spec.opt.learning_rate == 1e-4 and a.slots[1].trial_index == 51


def test_search_through_the_executor_uses_the_program_engine_and_adapts_its_limits():
    import mitune
    from mitune import tune
    from mitune.schedulers import ASHAScheduler
    tr = _trainable(engine="auto", capacity=6, model_kwargs={"max_d_model": 64, "max_layers": 2, "max_dim_feedforward": 128,
                                                             "max_batch": 16})
    space = dict(BASE, d_model=tune.choice([64, 128, 192]), num_heads=tune.choice([2, 4, 32]),
                 num_encoder_layers=tune.choice([1, 2, 4]), batch_size=tune.choice([16, 32]),
                 dim_feedforward=tune.sample_from(lambda spec: spec.config.d_model * tune.choice([2, 3])),
                 attention_type=tune.choice(["scaled_dot_product", "linear_attention", "multi_head_attention"]),
                 key_dim_scaling=tune.choice([1.0, 0.5, 0.25]), attn_kernel_size=tune.choice([3, 5, 7]),
                 depthwise_separable_conv=tune.choice([True, False]), shared_weights=tune.choice([True, False]),
                 dropout=tune.uniform(0.0, 0.3), stochastic_depth_rate=tune.uniform(0.0, 0.2),
                 learning_rate=tune.loguniform(1e-4, 1e-2), loss_function=tune.choice(["mse", "mae", "huber", "mape"]),
                 optimizer=tune.choice(["adam", "adamw", "sgd", "rmsprop"]), gradient_clipping=tune.uniform(0.0, 1.0),
                 warmup_steps=tune.choice([2, 8]), total_steps=tune.choice([50, 100]))
    an = mitune.run(tr, config=space, num_samples=14, scheduler=ASHAScheduler(metric="validation_mape", mode="min", max_t=3),
                    local_dir=None, verbose=0, seed=1, metric="validation_mape", mode="min", device=torch.device("cuda"))
    assert an.stats["engine"] == "xf" and "engine_fallback" not in an.stats
    assert len(an.trials) == 14 and all(t.status == "TERMINATED" for t in an.trials)
    assert all(np.isfinite(t.last_result["validation_mape"]) for t in an.trials)
    assert an.best_config is not None


def test_a_space_outside_the_program_engine_runs_on_the_per_slot_engine():
    import mitune
    from mitune import tune
    from mitune.schedulers import ASHAScheduler
    tr = _trainable(engine="auto", capacity=3)
    space = dict(BASE, d_model=tune.choice([64, 66]), num_heads=2, num_encoder_layers=1, dim_feedforward=128)     # 66 / 2 = 33-wide heads: fine; nq = 2*2*33+66 = 198: not a multiple of 4
    an = mitune.run(tr, config=space, num_samples=3, scheduler=ASHAScheduler(metric="validation_mape", mode="min", max_t=2),
                    local_dir=None, verbose=0, seed=1, metric="validation_mape", mode="min", device=torch.device("cuda"))
    assert an.stats["engine"] == "graph" and "engine_fallback" in an.stats
    assert all(np.isfinite(t.last_result["validation_mape"]) for t in an.trials)


def test_bf16_products_track_the_f32_pack_and_keep_packing_invariance():
    """``dtype="bf16"``: same parameters, activations and accumulation in f32, operands of the grouped GEMMs rounded to bf16
    for the matrix instruction.  Losses follow the f32 pack within bf16 rounding, are not identical to it (the switch is in
    force), and a trial's numbers still do not depend on its slot or neighbours."""
    import dataclasses
    tr32 = _trainable()
    tr16 = dataclasses.replace(tr32, dtype="bf16")
    packs = {}
    for name, tr in (("f32", tr32), ("bf16", tr16)):
        pack = _xf(tr)
        for slot, (i, c) in zip((5, 0, 3, 2, 7, 1), enumerate(CONFIGS)):
            pack.load(slot, 30 + i, c)
        for _ in range(2):
            pack.train_epoch()
        packs[name] = (pack, pack.evaluate())
    differs = 0
    for slot in (5, 0, 3, 2, 7, 1):
        a, b = packs["f32"][1][slot], packs["bf16"][1][slot]
        for key in ("train_loss", "validation_loss", "validation_mape"):
            assert np.isfinite(b[key])
            assert abs(a[key] - b[key]) <= 0.08 * abs(a[key]) + 2e-3, (slot, key, a[key], b[key])
        differs += a["validation_loss"] != b["validation_loss"]
    assert differs >= 4
    alone = _xf(tr16)
    alone.load(4, 30 + 2, CONFIGS[2])
    for _ in range(2):
        alone.train_epoch()
    assert alone.evaluate()[4] == packs["bf16"][1][3]
    wa, wp = alone.export(4)["model"], packs["bf16"][0].export(3)["model"]
    for k in wa:
        assert torch.equal(wa[k], wp[k]), k
