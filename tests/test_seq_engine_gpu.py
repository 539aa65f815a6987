"""The sequence-pack engine (grouped launches for small Transformer-encoder trials) against independent trainings.

Oracle: ``TorchPackEngine`` on the CPU -- one ordinary ``SimpleTransformerModel`` + ``torch.optim`` optimiser per
trial, plain f32 PyTorch ops -- fed the same initial weights and the same keyed-permutation batches.  The grouped
engine's dense layers are exact f32; its attention kernel rounds Q/K/V to bf16, which bounds the agreement at about
1e-2 relative.  Packing invariance (a trial's numbers do not depend on its slot or neighbours) is checked bitwise.
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


def test_one_sgd_step_gives_the_autograd_gradient_of_every_parameter():
    """lr = 1, no momentum, no decay: parameters before minus after one step are the gradients the engine computed."""
    from mitune.models.transformer import SimpleTransformerModel
    from mitune.packed.permutation import permute, round_keys
    tr = _trainable(n_train=32, n_val=8, model_kwargs={"optimizer": "sgd", "momentum": 0.0})
    eng = _seq(tr)
    cfgs = [dict(c, learning_rate=1.0, weight_decay=0.0) for c in CONFIGS]
    for i, c in enumerate(cfgs):
        eng.load(i, 10 + i, c)
    before = [eng.export(i)["model"] for i in range(len(cfgs))]
    eng.train_epoch()
    torch.cuda.synchronize()
    after = [eng.export(i)["model"] for i in range(len(cfgs))]
    X, y = tr.train
    for i, c in enumerate(cfgs):
        model = SimpleTransformerModel.from_config(c, F).double()
        model.load_state_dict({k: v.double() for k, v in before[i].items()})
        model.train()
        idx = torch.from_numpy(permute(np.arange(32, dtype=np.uint32), 32, round_keys(3, 10 + i, 0)))
        loss = torch.nn.functional.mse_loss(model(X[idx].double()).reshape(-1), y[idx].double())
        loss.backward()
        for name, p in model.named_parameters():
            got = (before[i][name] - after[i][name]).double().reshape(p.shape)
            ref = p.grad
            scale = ref.abs().max().item() + 1e-6
            err = (got - ref).abs().max().item() / scale
            fro = ((got - ref).norm() / (ref.norm() + 1e-9)).item()
            # the attention kernel's bf16 operands bound the agreement; three layers compound it
            assert err < 8e-2 and fro < 4e-2, f"config {i} {name}: gradient error max {err:.3e} / norm {fro:.3e} (|g| max {scale:.3e})"


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
            assert got[i][key] == pytest.approx(want[i][key], rel=3e-2, abs=2e-4), (i, key, got[i], want[i])
        a, b = eng.export(i)["model"], ref.export(i)["model"]
        for k in b:
            if k.endswith("qkv_proj.bias"):
                continue      # the key bias has a mathematically zero gradient (softmax is shift-invariant): Adam turns
                              # its rounding noise into +-lr steps of arbitrary sign
            assert torch.allclose(a[k].reshape(b[k].shape), b[k], atol=1.5e-2, rtol=5e-2), (i, k)


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
    from mitune.packed import engine_seq
    tr = _trainable()
    cfgs = [dict(c, dropout=0.2) for c in CONFIGS] + [dict(CONFIGS[1], dropout=0.1, learning_rate=2e-3)]
    results = {}
    for chains in (1, 2, 3):
        monkeypatch.setattr(engine_seq, "SEQ_CHAIN_RULE", [(2, chains)])
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
