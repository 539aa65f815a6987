"""Resident epochs (one launch per epoch, groups of workgroups training their own trials) against the grouped launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(n=6000, f=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(n, f, generator=g)
    w = torch.randn(f, generator=g)
    y = torch.tanh(X @ w * 0.3) * 2 + 0.3 * X[:, 0] * X[:, 1] + 0.05 * torch.randn(n, generator=g)
    k = int(0.8 * n)
    return (X[:k], y[:k]), (X[k:], y[k:])


# different depths, widths off the tile grid, four optimiser families, three losses, three batch sizes (= three step counts)
CONFIGS = [
    {"hidden_sizes": [64, 64], "batch_size": 128, "learning_rate": 3e-3, "optimizer": "adamw", "weight_decay": 1e-3},
    {"hidden_sizes": [128, 64, 32], "batch_size": 256, "learning_rate": 1e-3, "optimizer": "adam", "loss_function": "huber"},
    {"hidden_sizes": [40], "batch_size": 64, "learning_rate": 1e-2, "optimizer": "sgd", "momentum": 0.9},
    {"hidden_sizes": [256, 200], "batch_size": 512, "learning_rate": 2e-3, "optimizer": "rmsprop", "loss_function": "mae",
     "warmup_steps": 5, "total_steps": 200},
    {"hidden_sizes": [200, 136, 72], "batch_size": 128, "learning_rate": 2e-3, "optimizer": "adamw", "loss_function": "mape"},
]


def _run(trainable, configs, slots, epochs, **kw):
    from mitune.packed.engine_hip import HipPackEngine
    eng = HipPackEngine(trainable, **kw)
    for slot, (idx, cfg) in zip(slots, configs):
        eng.load(slot, idx, cfg)
    hist = []
    for _ in range(epochs):
        eng.train_epoch()
        hist.append(eng.evaluate())
    states = {slot: eng.export(slot) for slot in slots}
    n_res = eng.resident_epochs
    eng.close()
    return hist, states, n_res


def _trainable(epochs=2):
    from mitune.packed import PackedTrainable
    train, val = _data(seed=4)
    return PackedTrainable(train=train, val=val, max_epochs=epochs, val_chunk=512, dtype="f32",
                           model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})


def _same(a, b, slots):
    for slot in slots:
        for part in ("model",):
            for k, v in a[slot][part].items():
                assert torch.equal(v, b[slot][part][k]), (slot, k)
        for k in ("exp_avg", "exp_avg_sq"):
            for name, v in a[slot]["optimizer"][k].items():
                assert torch.equal(v, b[slot]["optimizer"][k][name]), (slot, k, name)


def test_resident_epochs_equal_the_grouped_launches_bit_for_bit(monkeypatch):
    """Same tile code, same descriptor tables, unsplit weight gradients on both sides: parameters, both optimiser moments
    and every reported metric agree exactly."""
    from mitune.packed import engine_hip
    tr = _trainable()
    cfgs, slots = list(enumerate(CONFIGS)), [4, 0, 3, 1, 6]
    monkeypatch.setattr(engine_hip, "KSPLITS_SMALL", 1)
    monkeypatch.setattr(engine_hip, "RESIDENT", "0")
    h0, s0, n0 = _run(tr, cfgs, slots, 2, device="cuda", capacity=8, seed=5)
    monkeypatch.setattr(engine_hip, "RESIDENT", "auto")
    h1, s1, n1 = _run(tr, cfgs, slots, 2, device="cuda", capacity=8, seed=5)
    assert n0 == 0 and n1 == 2
    _same(s0, s1, slots)
    assert h0 == h1


@pytest.mark.parametrize("edge,gmin", [(64, 8), (32, 8), (64, 200), (32, 300)])
def test_resident_results_do_not_depend_on_tile_edge_or_grouping(monkeypatch, edge, gmin):
    """Tile edge 32 or 64; a group per trial, or (few, large groups) several trials trained one after the other by the
    same group: a trial's numbers are the same, and the same as when it trains alone."""
    from mitune.packed import engine_hip
    tr = _trainable()
    cfgs, slots = list(enumerate(CONFIGS)), [4, 0, 3, 1, 6]
    monkeypatch.setattr(engine_hip, "RESIDENT", "auto")
    monkeypatch.setattr(engine_hip, "RES_TILE", 64)
    monkeypatch.setattr(engine_hip, "RES_GMIN", 8)
    h0, s0, _ = _run(tr, cfgs, slots, 2, device="cuda", capacity=8, seed=5)
    monkeypatch.setattr(engine_hip, "RES_TILE", edge)
    monkeypatch.setattr(engine_hip, "RES_GMIN", gmin)
    h1, s1, n1 = _run(tr, cfgs, slots, 2, device="cuda", capacity=8, seed=5)
    assert n1 == 2
    _same(s0, s1, slots)
    assert h0 == h1
    _, alone, _ = _run(tr, [(1, CONFIGS[1])], [2], 2, device="cuda", capacity=8, seed=5)
    for k, v in alone[2]["model"].items():
        assert torch.equal(v, s1[0]["model"][k]), k


def test_packs_the_resident_kernel_does_not_cover_fall_back(monkeypatch):
    from mitune.packed import engine_hip
    tr = _trainable()
    monkeypatch.setattr(engine_hip, "RESIDENT", "auto")
    clipped = dict(CONFIGS[0], gradient_clipping=1.0)
    _, _, n = _run(tr, [(0, clipped), (1, CONFIGS[1])], [0, 1], 1, device="cuda", capacity=4, seed=5)
    assert n == 0
    monkeypatch.setattr(engine_hip, "RES_MAX_SLOTS", 1)
    _, _, n = _run(tr, [(0, CONFIGS[0]), (1, CONFIGS[1])], [0, 1], 1, device="cuda", capacity=4, seed=5)
    assert n == 0


def test_membership_changes_between_resident_epochs(monkeypatch):
    """Evict one trial and load another between epochs: the launch state is rebuilt, survivors continue unchanged."""
    from mitune.packed import engine_hip
    from mitune.packed.engine_hip import HipPackEngine
    monkeypatch.setattr(engine_hip, "RESIDENT", "auto")
    tr = _trainable(epochs=3)

    def go(swap):
        eng = HipPackEngine(tr, device="cuda", capacity=4, seed=5)
        eng.load(0, 0, CONFIGS[0])
        eng.load(1, 1, CONFIGS[1])
        eng.train_epoch()
        eng.evaluate()
        if swap:
            eng.evict(0)
            eng.load(3, 4, CONFIGS[4])
        eng.train_epoch()
        out = eng.export(1)
        assert eng.resident_epochs == 2
        eng.close()
        return out

    a, b = go(False), go(True)
    for k, v in a["model"].items():
        assert torch.equal(v, b["model"][k]), k
