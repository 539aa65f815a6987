"""The sequence-pack engine (grouped launches for small Transformer-encoder trials) against independent trainings.

Oracles: (a) a float64 re-statement of the model in this file that replays the kernels' counter-based dropout masks, for
the gradients of one step (every product of the engine is f32 -- exact-f32 MFMA in the dense layers, f32 multiply-add
chains in attention -- so they agree to about 1e-5 of the largest gradient, dropout on or off); (b) ``TorchPackEngine``
on the CPU -- one ordinary ``SimpleTransformerModel`` + ``torch.optim`` optimiser per trial, plain f32 PyTorch ops -- fed
the same initial weights and the same keyed-permutation batches, for whole epochs.  The attention form that rounds
Q / K / V to bf16 for the matrix instruction (``attn_exact = 0``) is held to its own, looser bound.  Packing invariance
(a trial's numbers do not depend on its slot or neighbours) is checked bitwise.
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, F = 50, 10
CONFIGS = [
    {"d_model": 32, "num_heads": 2, "num_layers": 1, "dropout": 0.0, "learning_rate": 2e-3, "weight_decay": 1e-4},
    {"d_model": 64, "num_heads": 4, "num_layers": 2, "dropout": 0.0, "learning_rate": 1e-3, "weight_decay": 1e-3},
    {"d_model": 128, "num_heads": 2, "num_layers": 3, "dropout": 0.0, "learning_rate": 5e-4, "weight_decay": 1e-5},
    {"d_model": 32, "num_heads": 4, "num_layers": 3, "dropout": 0.0, "learning_rate": 3e-3, "weight_decay": 1e-6},
]


def _data(n_train=256, n_val=72, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(n_train + n_val, S, F, generator=g)
    w = torch.randn(F, generator=g) / F ** 0.5
    y = torch.tanh((X[:, -5:, :] @ w).mean(1)) + 0.05 * torch.randn(n_train + n_val, generator=g)
    return (X[:n_train], y[:n_train]), (X[n_train:], y[n_train:])


def _trainable(engine="seq", **kw):
    from mitune.packed import PackedTrainable
    train, val = _data(kw.pop("n_train", 256), kw.pop("n_val", 72))
    mk = {"batch_size": 32, "optimizer": "adamw"}
    mk.update(kw.pop("model_kwargs", {}))
    return PackedTrainable(model="simple_transformer", train=train, val=val, max_epochs=4, dtype="f32", engine=engine,
                           capacity=kw.pop("capacity", 8), model_kwargs=mk, **kw)


def _seq(tr, seed=3):
    from mitune.packed.engine_seq import SeqPackEngine
    return SeqPackEngine(tr, "cuda", tr.capacity, seed)


def test_layout_covers_every_parameter_of_the_module():
    from mitune.models.transformer import SimpleTransformerModel
    from mitune.packed.engine_seq import SeqLayout
    for cfg in CONFIGS:
        model = SimpleTransformerModel.from_config(cfg, F)
        lay = SeqLayout(F, 12, cfg["d_model"], cfg["num_layers"])
        keys = {k for k, _, _, _ in lay.entries}
        assert keys == set(model.state_dict().keys())
        for key, off, (rows, cols), ld in lay.entries:
            assert model.state_dict()[key].numel() == rows * cols and off % 8 == 0 and ld >= cols
        ends = sorted((off, off + rows * ld) for _, off, (rows, _), ld in lay.entries)
        assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= lay.total


def _keep(seed_words, site, step, index, p):
    """Keep flags of the kernels' counter-based dropout (csrc/mt_common.h: mt_site_seed + mt_rng_hash): element ``index``
    of dropout site ``site`` at step ``step`` is kept iff its hash is at least p * 2^32."""
    from mitune.packed.permutation import mix32
    u = np.uint32
    with np.errstate(over="ignore"):
        s0 = u(seed_words[0]) + u(site) * u(0x9E3779B9)
        s1 = u(seed_words[1]) ^ mix32(u(step) * u(0x85EBCA6B) + u(site) + u(0x51ED270B))
        i = np.asarray(index).astype(np.uint32)
        h = mix32((i * u(0x9E3779B9) + s0) ^ s1)
    return torch.from_numpy(h >= u(min(int(p * 4294967296.0), 0xFFFFFFFF)))


def _oracle_loss(w, X, y, cfg, seed_words, step):
    """MSE of the sample Transformer on one batch in float64, written out operation by operation, with the dropout masks
    the grouped kernels draw at (seed words of the slot's epoch, site 4 l + {0: attention probabilities, 1: attention
    branch, 2: feed-forward hidden, 3: feed-forward branch}, step, element index)."""
    d, H, layers, p = cfg["d_model"], cfg["num_heads"], cfg["num_layers"], cfg["dropout"]
    B, S, _ = X.shape
    dk, m, ks = d // H, B * S, 1.0 / (1.0 - p)
    ln = torch.nn.functional.layer_norm

    def mask(site, shape):
        if p <= 0:
            return 1.0
        return _keep(seed_words, site, step, np.arange(int(np.prod(shape)), dtype=np.int64), p).reshape(shape).double() * ks

    x = X.reshape(m, -1) @ w["input_projection.weight"].t() + w["input_projection.bias"]
    for l in range(layers):
        pre = f"transformer.{l}."
        qkv = (x @ w[pre + "self_attn.qkv_proj.weight"].t() + w[pre + "self_attn.qkv_proj.bias"]).reshape(B, S, 3, H, dk)
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))                 # [B, H, S, dk]
        prob = torch.softmax(q @ k.transpose(-1, -2) / dk ** 0.5, dim=-1) * mask(4 * l, (B, H, S, S))
        o = (prob @ v).permute(0, 2, 1, 3).reshape(m, d)
        a = o @ w[pre + "self_attn.out_proj.weight"].t() + w[pre + "self_attn.out_proj.bias"]
        h1 = ln(x + a * mask(4 * l + 1, (m, d)), (d,), w[pre + "norm1.weight"], w[pre + "norm1.bias"], 1e-5)
        f = torch.relu(h1 @ w[pre + "linear1.weight"].t() + w[pre + "linear1.bias"]) * mask(4 * l + 2, (m, 4 * d))
        f = f @ w[pre + "linear2.weight"].t() + w[pre + "linear2.bias"]
        x = ln(h1 + f * mask(4 * l + 3, (m, d)), (d,), w[pre + "norm2.weight"], w[pre + "norm2.bias"], 1e-5)
    out = x.reshape(B, S, d)[:, -1, :] @ w["fc.weight"].t() + w["fc.bias"]
    return torch.nn.functional.mse_loss(out.reshape(-1), y)


def _one_sgd_step_gradients(dropout):
    """lr = 1, no momentum, no decay: parameters before minus after one step are the gradients the engine computed.
    Returns [(config index, parameter name, engine gradient, float64 oracle gradient)]."""
    from mitune.packed.permutation import permute, round_keys
    tr = _trainable(n_train=32, n_val=8, model_kwargs={"optimizer": "sgd", "momentum": 0.0})
    eng = _seq(tr)
    cfgs = [dict(c, learning_rate=1.0, weight_decay=0.0, dropout=dropout) for c in CONFIGS]
    for i, c in enumerate(cfgs):
        eng.load(i, 10 + i, c)
    before = [eng.export(i)["model"] for i in range(len(cfgs))]
    seeds = [eng._slot_seed(eng.slots[i]) for i in range(len(cfgs))]
    eng.train_epoch()
    torch.cuda.synchronize()
    after = [eng.export(i)["model"] for i in range(len(cfgs))]
    X, y = tr.train
    out = []
    for i, c in enumerate(cfgs):
        w = {k: v.double().requires_grad_(True) for k, v in before[i].items()}
        idx = torch.from_numpy(permute(np.arange(32, dtype=np.uint32), 32, round_keys(3, 10 + i, 0)).astype(np.int64))
        _oracle_loss(w, X[idx].double(), y[idx].double(), c, seeds[i], 0).backward()
        for name, t in w.items():
            out.append((i, name, (before[i][name] - after[i][name]).double().reshape(t.shape), t.grad))
    return out


@pytest.mark.parametrize("dropout", [0.0, 0.25])
def test_one_sgd_step_gives_the_float64_gradient_of_every_parameter(dropout):
    """Exact-f32 path (the default), dropout off and on: 1e-4 of the largest gradient element, per parameter.  A wrong
    keep scale, a mask drawn from the wrong site / step / index, or bf16-rounded attention operands all fail this."""
    for i, name, got, ref in _one_sgd_step_gradients(dropout):
        scale = ref.abs().max().item() + 1e-9
        err = (got - ref).abs().max().item()
        assert err <= 1e-4 * scale + 1e-7, f"config {i} {name}: |got - ref| max {err:.3e}, |ref| max {scale:.3e}"


def test_matrix_form_attention_keeps_its_own_looser_bound():
    """``attn_exact = 0``: Q / K / V are rounded to bf16 for the matrix instruction; three layers compound the rounding."""
    from mitune.ops import lib as L
    so = L.load()
    was = so.mt_attn_exact_level(0)
    try:
        grads = _one_sgd_step_gradients(0.0)
    finally:
        so.mt_attn_exact_level(was)
    worst = 0.0
    for i, name, got, ref in grads:
        scale = ref.abs().max().item() + 1e-6
        err = (got - ref).abs().max().item() / scale
        fro = ((got - ref).norm() / (ref.norm() + 1e-9)).item()
        worst = max(worst, err)
        assert err < 8e-2 and fro < 4e-2, f"config {i} {name}: gradient error max {err:.3e} / norm {fro:.3e}"
    assert worst > 1e-4          # the rounding is visible: this really was the other form


def test_pack_matches_independent_eager_trainings():
    from mitune.packed.engine import TorchPackEngine
    tr = _trainable()
    eng = _seq(tr)
    ref = TorchPackEngine(copy.copy(tr), "cpu", tr.capacity, 3)
    for i, c in enumerate(CONFIGS):
        eng.load(i, 20 + i, c)
        ref.load(i, 20 + i, c)
    for _ in range(2):
        eng.train_epoch()
        ref.train_epoch()
    got, want = eng.evaluate(), ref.evaluate()
    for i in range(len(CONFIGS)):
        for key in ("train_loss", "validation_mse", "validation_loss"):
            assert got[i][key] == pytest.approx(want[i][key], rel=2e-3, abs=2e-5), (i, key, got[i], want[i])
        a, b = eng.export(i)["model"], ref.export(i)["model"]
        for k in b:
            if k.endswith("qkv_proj.bias"):
                continue      # the key bias has a mathematically zero gradient (softmax is shift-invariant): Adam turns
                              # its rounding noise into +-lr steps of arbitrary sign
            assert torch.allclose(a[k].reshape(b[k].shape), b[k], atol=2e-3, rtol=2e-2), (i, k)


def test_a_trial_does_not_depend_on_its_slot_or_neighbours():
    tr = _trainable()
    cfgs = [dict(c, dropout=0.3) for c in CONFIGS]
    alone = _seq(tr)
    alone.load(0, 31, cfgs[1])
    for _ in range(2):
        alone.train_epoch()
    m_alone, w_alone = alone.evaluate()[0], alone.export(0)["model"]
    pack = _seq(tr)
    for slot, (idx, c) in zip((5, 0, 3, 2), ((30, cfgs[0]), (31, cfgs[1]), (32, cfgs[2]), (33, cfgs[3]))):
        pack.load(slot, idx, c)
    for _ in range(2):
        pack.train_epoch()
    m_pack, w_pack = pack.evaluate()[0], pack.export(0)["model"]
    assert m_alone == m_pack
    for k in w_alone:
        assert torch.equal(w_alone[k], w_pack[k]), k


def test_dropout_trains_and_validation_is_deterministic():
    tr = _trainable()
    eng = _seq(tr)
    eng.load(0, 40, dict(CONFIGS[1], dropout=0.5))
    eng.load(1, 40, dict(CONFIGS[1], dropout=0.0))
    eng.train_epoch()
    a, b = eng.evaluate(), eng.evaluate()
    assert a == b
    assert np.isfinite(a[0]["train_loss"]) and np.isfinite(a[0]["validation_mse"])
    assert a[0]["train_loss"] != a[1]["train_loss"]            # same trial stream, dropout changes the trajectory
    w0, w1 = eng.export(0)["model"], eng.export(1)["model"]
    assert not torch.equal(w0["fc.weight"], w1["fc.weight"])


def test_export_load_round_trip_continues_like_an_uninterrupted_run():
    tr = _trainable()
    a = _seq(tr)
    a.load(0, 50, dict(CONFIGS[2], dropout=0.2))
    a.train_epoch()
    state = a.export(0)
    a.train_epoch()
    want = a.evaluate()[0]
    b = _seq(tr)
    b.load(4, 50, dict(CONFIGS[2], dropout=0.2), state=state)
    b.train_epoch()
    got = b.evaluate()[4]
    assert got == want
    # the exported weights load into the plain module
    from mitune.models.transformer import SimpleTransformerModel
    SimpleTransformerModel.from_config(CONFIGS[2], F).load_state_dict(state["model"])


def test_clone_copies_weights_and_moments_inside_the_pack():
    tr = _trainable()
    eng = _seq(tr)
    eng.load(0, 60, CONFIGS[0])
    eng.load(1, 61, CONFIGS[0])
    eng.train_epoch()
    eng.clone(0, 1, dict(CONFIGS[0], learning_rate=1e-4))
    a, b = eng.export(0), eng.export(1)
    for k in a["model"]:
        assert torch.equal(a["model"][k], b["model"][k])
        assert torch.equal(a["optimizer"]["exp_avg"][k], b["optimizer"]["exp_avg"][k])
    assert eng.slots[1].spec.opt.learning_rate == 1e-4 and eng.slots[1].trial_index == 61


def test_search_through_the_executor_uses_the_grouped_engine():
    import mitune
    from mitune.tune.schedulers import ASHAScheduler
    from mitune import tune
    tr = _trainable(engine="auto", capacity=6)
    space = {"d_model": tune.choice([32, 64, 128]), "num_heads": tune.choice([2, 4]), "num_layers": tune.choice([1, 2, 3]),
             "dropout": tune.uniform(0.1, 0.5), "learning_rate": tune.loguniform(1e-4, 1e-2),
             "weight_decay": tune.loguniform(1e-6, 1e-2)}
    an = mitune.run(tr, config=space, num_samples=10, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=4),
                    local_dir=None, verbose=0, seed=1, metric="validation_loss", mode="min", device=torch.device("cuda"))
    assert an.stats["engine"] == "seq"
    assert len(an.trials) == 10 and all(np.isfinite(t.last_result["validation_loss"]) for t in an.trials)
    assert an.best_config is not None


def test_unsupported_shapes_fall_back_to_the_per_slot_engine():
    from mitune.packed import engine_seq
    tr = _trainable(engine="auto")
    assert engine_seq.supports(tr, {"d_model": 48, "num_heads": 4, "num_layers": 1}) is not None       # heads 12 wide
    assert engine_seq.supports(tr, {"d_model": 64, "num_heads": 4, "num_layers": 2}) is None


def test_a_search_space_outside_the_grouped_kernels_runs_on_the_per_slot_engine():
    """d_model 48 with 4 heads (12-wide heads, not a multiple of 8) is in the space: the executor checks every
    combination before the first trial and takes the one-graph-per-slot engine; a space inside the envelope but past the
    default limits (batch 64) raises the limits and stays on grouped launches."""
    import mitune
    from mitune import tune
    from mitune.schedulers import ASHAScheduler
    space = {"d_model": tune.choice([32, 48]), "num_heads": tune.choice([2, 4]), "num_layers": 1, "dropout": 0.1,
             "learning_rate": tune.loguniform(1e-4, 1e-2), "weight_decay": 1e-4}
    an = mitune.run(_trainable(engine="auto", capacity=4), config=space, num_samples=4, local_dir=None, verbose=0, seed=2,
                    scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=2), device=torch.device("cuda"))
    assert an.stats["engine"] == "graph" and "engine_fallback" in an.stats
    assert all(np.isfinite(t.last_result["validation_loss"]) for t in an.trials)
    space = {"d_model": 32, "num_heads": 2, "num_layers": tune.choice([1, 2]), "dropout": 0.1, "batch_size": tune.choice([32, 64]),
             "learning_rate": tune.loguniform(1e-4, 1e-2), "weight_decay": 1e-4}
    an = mitune.run(_trainable(engine="auto", capacity=4), config=space, num_samples=4, local_dir=None, verbose=0, seed=2,
                    scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=2), device=torch.device("cuda"))
    assert an.stats["engine"] == "seq" and "engine_fallback" not in an.stats
    assert all(np.isfinite(t.last_result["validation_loss"]) for t in an.trials)


def test_two_ranks_move_sequence_trials_as_device_slabs_without_changing_them():
    """Two executors wired through threads, both packs on this GPU: every even trial stops after its first epoch, so one
    rank runs empty and trials move over as raw device slabs; no trial's numbers change against the run without moves."""
    import threading
    import mitune
    from mitune import tune
    from mitune.packed import PackedExecutor
    from mitune.parallel.comm import ThreadComm
    from mitune.schedulers import ASHAScheduler

    class StopEven(ASHAScheduler):
        def on_batch_result(self, runner, trials, results):
            return [self.STOP if (t.index % 2 == 0 or r["training_iteration"] >= 3) else self.CONTINUE
                    for t, r in zip(trials, results)]

    space = {"d_model": tune.choice([32, 64]), "num_heads": tune.choice([2, 4]), "num_layers": tune.choice([1, 2]),
             "dropout": tune.uniform(0.1, 0.3), "learning_rate": tune.loguniform(1e-4, 1e-2), "weight_decay": 1e-4}

    def search(gap):
        comms = ThreadComm.make(2)
        out, errs = [None, None], []

        def body(r):
            try:
                tr = _trainable(engine="auto", capacity=4, rebalance_gap=gap)
                ex = PackedExecutor(tr, comm=comms[r], device=torch.device("cuda", 0), capacity=4, seed=0)
                an = mitune.run(tr, config=space, num_samples=8, executor=ex, seed=0, verbose=0,
                                scheduler=StopEven(metric="validation_loss", mode="min", max_t=3))
                out[r] = ({t.trial_id: (t.training_iteration, t.last_result["validation_loss"], t.last_result["train_loss"])
                           for t in an.trials}, an.stats["migrations"], an.stats["engine"])
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
                comms[r].hub.barrier.abort()

        ths = [threading.Thread(target=body, args=(r,)) for r in range(2)]
        [t.start() for t in ths]
        [t.join(timeout=100) for t in ths]
        if errs:
            raise errs[0]
        assert out[0] == out[1]
        return out[0]

    res, moves, engine = search(2)
    res0, moves0, _ = search(0)
    assert engine == "seq" and moves >= 1 and moves0 == 0
    assert res == res0


def test_chains_on_separate_streams_change_nothing(monkeypatch):
    """A pack split into independent chains of launches on two and three HIP streams: bitwise the one-chain results."""
    from mitune.options import OPTIONS
    tr = _trainable()
    cfgs = [dict(c, dropout=0.2) for c in CONFIGS] + [dict(CONFIGS[1], dropout=0.1, learning_rate=2e-3)]
    results = {}
    for chains in (1, 2, 3):
        monkeypatch.setattr(OPTIONS, "seq_chain_rule", f"2:{chains}")
        eng = _seq(tr)
        for i, c in enumerate(cfgs):
            eng.load(i, 70 + i, c)
        for _ in range(2):
            eng.train_epoch()
        assert len(eng._chains) == chains
        results[chains] = (eng.evaluate(), [eng.export(i)["model"] for i in range(len(cfgs))])
    for chains in (2, 3):
        assert results[chains][0] == results[1][0]
        for a, b in zip(results[chains][1], results[1][1]):
            for k in a:
                assert torch.equal(a[k], b[k]), (chains, k)
