"""The HIP pack engine against the per-trial PyTorch oracle, and packing invariance."""
import numpy as np
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


CONFIGS = [
    {"hidden_sizes": [64, 64], "batch_size": 128, "learning_rate": 3e-3, "optimizer": "adamw", "weight_decay": 1e-3},
    {"hidden_sizes": [128, 64, 32], "batch_size": 256, "learning_rate": 1e-3, "optimizer": "adam", "loss_function": "huber"},
    {"hidden_sizes": [40], "batch_size": 64, "learning_rate": 1e-2, "optimizer": "sgd", "momentum": 0.9,
     "gradient_clipping": 1.0},
    {"hidden_sizes": [256, 200], "batch_size": 512, "learning_rate": 2e-3, "optimizer": "rmsprop", "loss_function": "mae",
     "warmup_steps": 5, "total_steps": 200},
]


def _run(engine_cls, trainable, configs, slots, epochs, **kw):
    eng = engine_cls(trainable, **kw)
    for slot, (idx, cfg) in zip(slots, configs):
        eng.load(slot, idx, cfg)
    hist = []
    for _ in range(epochs):
        eng.train_epoch()
        hist.append(eng.evaluate())
    states = {slot: eng.export(slot) for slot in slots}
    return hist, states


@pytest.mark.parametrize("dtype,rel,abs_frac", [("bf16", 0.08, 0.02), ("f32", 1e-3, 1e-5)])
def test_hip_engine_tracks_torch_oracle(dtype, rel, abs_frac):
    """Three epochs of four heterogeneous trials: bf16 MFMA stays within bf16 rounding of the f32 oracle; the exact-f32
    engine (same arithmetic as the oracle, only the summation order differs) within 1e-3 relative."""
    from mitune.packed import PackedTrainable, TorchPackEngine
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data()
    tr = PackedTrainable(train=train, val=val, max_epochs=3, val_chunk=512, dtype=dtype,
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    cfgs = list(enumerate(CONFIGS))
    slots = [0, 1, 2, 3]
    h_hip, s_hip = _run(HipPackEngine, tr, cfgs, slots, 3, device="cuda", capacity=6, seed=3)
    h_ref, s_ref = _run(TorchPackEngine, tr, cfgs, slots, 3, device="cuda", capacity=6, seed=3)
    var_y = float(val[1].var())
    for e in range(3):
        for slot in slots:
            a, b = h_hip[e][slot], h_ref[e][slot]
            for k in ("train_loss", "validation_loss", "validation_mse"):
                assert a[k] == pytest.approx(b[k], rel=rel, abs=abs_frac * var_y), (e, slot, k, a, b)
    # the model actually learns (bf16 compute does not stall training)
    assert h_hip[-1][0]["validation_mse"] < 0.5 * var_y
    # exported checkpoint loads into the plain single-trial module
    from mitune.models.mlp import MLPRegressor
    m = MLPRegressor(train[0].shape[1], CONFIGS[1]["hidden_sizes"])
    m.load_state_dict(s_hip[1]["model"])
    with torch.no_grad():
        mse = float(((m(val[0]) - val[1]) ** 2).mean())
    assert mse == pytest.approx(h_hip[-1][1]["validation_mse"], rel=0.05)


def test_unknown_dtype_is_refused():
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(n=500)
    tr = PackedTrainable(train=train, val=val, dtype="fp8")
    with pytest.raises(ValueError, match="dtype"):
        HipPackEngine(tr, device="cuda", capacity=2)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_packing_invariance_is_bitwise(dtype):
    """A trial's weights do not depend on its slot or on which other trials share the pack."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=1)
    tr = PackedTrainable(train=train, val=val, max_epochs=2, val_chunk=512, dtype=dtype,
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    _, alone = _run(HipPackEngine, tr, [(7, CONFIGS[1])], [0], 2, device="cuda", capacity=8, seed=5)
    _, packed = _run(HipPackEngine, tr, [(3, CONFIGS[0]), (7, CONFIGS[1]), (9, CONFIGS[3])], [5, 2, 6], 2,
                     device="cuda", capacity=8, seed=5)
    for k, v in alone[0]["model"].items():
        assert torch.equal(v, packed[2]["model"][k]), k


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_a_small_trial_alone_equals_the_same_trial_among_wide_neighbours(dtype):
    """A pack of one narrow trial is a handful of weight-gradient tiles, a pack with six wide neighbours a hundred: launch
    forms, tile edges and (by default) split counts may follow that, the trial's weights may not."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=2)
    tr = PackedTrainable(train=train, val=val, max_epochs=2, val_chunk=512, dtype=dtype,
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    _, alone = _run(HipPackEngine, tr, [(7, CONFIGS[1])], [0], 2, device="cuda", capacity=8, seed=5)
    crowd = [(7, CONFIGS[1])] + [(20 + i, dict(CONFIGS[3], learning_rate=1e-3 * (i + 1))) for i in range(6)]
    _, packed = _run(HipPackEngine, tr, crowd, [3, 0, 1, 2, 4, 5, 6], 2, device="cuda", capacity=8, seed=5)
    for k, v in alone[0]["model"].items():
        assert torch.equal(v, packed[3]["model"][k]), k


def test_f32_results_do_not_depend_on_the_tile_edge(monkeypatch):
    """The exact-f32 engine picks 128-, 64- or 32-tiles from the size of the pack; a trial's weights must not notice."""
    from mitune.options import OPTIONS
    from mitune.packed import PackedTrainable, engine_hip
    train, val = _data(seed=8)
    tr = PackedTrainable(train=train, val=val, max_epochs=2, val_chunk=512, dtype="f32",
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    out = {}
    for edge, (small, tiny) in {128: (0, 0), 64: (10 ** 9, 0), 32: (10 ** 9, 10 ** 9)}.items():
        monkeypatch.setattr(OPTIONS, "f32_small_tiles", small)
        monkeypatch.setattr(OPTIONS, "f32_tiny_tiles", tiny)
        _, out[edge] = _run(engine_hip.HipPackEngine, tr, [(7, CONFIGS[1]), (9, CONFIGS[3])], [0, 1], 2, device="cuda",
                            capacity=4, seed=5)
    for slot in (0, 1):
        for k, v in out[128][slot]["model"].items():
            assert torch.equal(v, out[64][slot]["model"][k]), (slot, k)
            assert torch.equal(v, out[32][slot]["model"][k]), (slot, k)


def test_a_slot_moved_between_engines_continues_bit_for_bit():
    """export_raw -> import_raw (what a trial's move to another GPU does) mid-training changes nothing."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=4)
    tr = PackedTrainable(train=train, val=val, max_epochs=4, val_chunk=512,
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    hist, stay = _run(HipPackEngine, tr, [(11, CONFIGS[1]), (12, CONFIGS[3])], [1, 4], 4, device="cuda", capacity=6, seed=2)
    a = HipPackEngine(tr, device="cuda", capacity=6, seed=2)
    b = HipPackEngine(tr, device="cuda", capacity=6, seed=2)
    a.load(1, 11, CONFIGS[1])
    a.load(4, 12, CONFIGS[3])
    b.load(0, 20, CONFIGS[0])                        # the destination pack has a tenant of its own
    for _ in range(2):
        a.train_epoch()
        b.train_epoch()
    raw = a.export_raw(4)
    assert raw.numel() == a.raw_numel and raw.device.type == "cuda"
    a.evict(4)
    with pytest.raises(ValueError):
        b.import_raw(3, 13, CONFIGS[3], raw)         # a slab is bound to its trial
    b.import_raw(3, 12, CONFIGS[3], raw)
    for _ in range(2):
        a.train_epoch()
        b.train_epoch()
    assert b.evaluate()[3] == hist[-1][4] and a.evaluate()[1] == hist[-1][1]
    moved = b.export(3)
    for k, v in stay[4]["model"].items():
        assert torch.equal(v, moved["model"][k]), k
    for k, v in stay[4]["optimizer"]["exp_avg_sq"].items():
        assert torch.equal(v, moved["optimizer"]["exp_avg_sq"][k]), k
    assert moved["epoch"] == 4 and moved["step"] == stay[4]["step"]


def test_two_ranks_rebalance_hip_packs_after_uneven_stopping():
    """Two executors (ranks wired through threads, both packs on this GPU): one rank loses all its trials after the
    first epoch, some of the other's move over as device slabs, and no trial's losses change."""
    import threading
    import mitune
    from mitune import tune
    from mitune.packed import PackedExecutor, PackedTrainable
    from mitune.parallel.comm import ThreadComm
    from mitune.schedulers import ASHAScheduler

    class StopEven(ASHAScheduler):
        def on_batch_result(self, runner, trials, results):
            return [self.STOP if (t.index % 2 == 0 or r["training_iteration"] >= 4) else self.CONTINUE
                    for t, r in zip(trials, results)]

    train, val = _data(n=8000, f=16, seed=6)
    space = {"width": tune.choice([32, 64, 128]), "num_layers": tune.choice([1, 2]),
             "learning_rate": tune.loguniform(1e-3, 1e-2), "batch_size": tune.choice([128, 256])}

    def search(gap):
        comms = ThreadComm.make(2)
        out, errs = [None, None], []

        def body(r):
            try:
                tr = PackedTrainable(train=train, val=val, max_epochs=4, val_chunk=2048, capacity=4, rebalance_gap=gap,
                                     model_kwargs={"max_width": 128, "max_layers": 2, "max_batch": 256})
                ex = PackedExecutor(tr, comm=comms[r], device=torch.device("cuda", 0), capacity=4, seed=0)
                an = mitune.run(tr, config=space, num_samples=8, executor=ex, seed=0, verbose=0,
                                scheduler=StopEven(metric="validation_loss", mode="min", max_t=4))
                out[r] = ({t.trial_id: (t.training_iteration, t.last_result["validation_loss"], t.last_result["train_loss"])
                           for t in an.trials},
                          sorted(t.placement["rank"] for t in an.trials if t.training_iteration == 4),
                          an.stats["migrations"], an.stats["engine"])
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

    res, ranks, moves, engine = search(2)
    res0, ranks0, moves0, _ = search(0)
    # the HIP engine says what each trial weighs (width, depth, batch size), so the ranks are levelled by weight, not by
    # head count: at least one of the four survivors moves, and both ranks hold some
    assert engine == "hip" and 1 <= moves <= 3 and set(ranks) == {0, 1}
    assert moves0 == 0 and len(ranks0) == 4 and ranks0 != ranks      # without moves they stay where they were dealt
    assert res == res0


def test_hip_packs_survive_a_vanished_rank():
    """Two executors with HIP packs (ranks wired through threads); rank 1 vanishes before its third epoch.  Rank 0 shrinks
    the group to itself, loads the lost trials into its own pack from their initial weights and finishes the search; what
    any trial reports at an iteration equals what the undisturbed one-rank search reports there, bit for bit."""
    import threading
    import mitune
    from mitune import tune
    from mitune.packed import PackedExecutor, PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    from mitune.parallel.comm import LocalComm, ThreadComm
    from mitune.schedulers import ASHAScheduler

    train, val = _data(n=8000, f=16, seed=6)
    space = {"width": tune.choice([32, 64, 128]), "num_layers": tune.choice([1, 2]),
             "learning_rate": tune.loguniform(1e-3, 1e-2), "batch_size": tune.choice([128, 256])}

    class Vanish(BaseException):
        pass

    def search(comm, dies):
        class Dying(HipPackEngine):
            epochs_done = 0

            def train_epoch(self):
                if dies and self.epochs_done == 2:
                    comm.hub.leave(comm.rank)
                    raise Vanish()
                super().train_epoch()
                self.epochs_done += 1

        tr = PackedTrainable(train=train, val=val, max_epochs=4, val_chunk=2048, capacity=8, dtype="f32",
                             model_kwargs={"max_width": 128, "max_layers": 2, "max_batch": 256})
        dev = torch.device("cuda", 0)
        ex = PackedExecutor(tr, comm=comm, device=dev, capacity=8, seed=0, engine=Dying(tr, dev, 8, 0))
        an = mitune.run(tr, config=space, num_samples=8, executor=ex, seed=0, verbose=0,
                        scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=4, reduction_factor=2))
        return ({t.trial_id: {r["training_iteration"]: (r["validation_loss"], r["train_loss"]) for r in t.results}
                 for t in an.trials}, {t.trial_id: t.status for t in an.trials}, dict(an.stats))

    ref, _, _ = search(LocalComm(torch.device("cuda", 0)), False)
    comms = ThreadComm.make(2)
    out, errs = {}, []

    def body(r):
        try:
            out[r] = search(comms[r], r == 1)
        except Vanish:
            pass
        except BaseException as e:      # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=body, args=(r,)) for r in range(2)]
    [t.start() for t in ths]
    [t.join(timeout=100) for t in ths]
    if errs:
        raise errs[0]
    assert sorted(out) == [0]
    traj, status, stats = out[0]
    assert len(status) == 8 and set(status.values()) == {"TERMINATED"}
    assert stats["ranks_lost"] == 1 and stats["trials_recovered"] >= 1 and stats["world_size"] == 1
    same = 0
    for tid, hist in traj.items():
        for it, v in hist.items():
            if it in ref[tid]:
                assert v == ref[tid][it], (tid, it)
                same += 1
    assert same >= 8


def test_end_to_end_packed_search_on_gpu(tmp_path):
    import mitune
    from mitune import tune
    from mitune.packed import PackedTrainable
    from mitune.schedulers import ASHAScheduler
    train, val = _data(n=20000, f=16, seed=2)
    tr = PackedTrainable(train=train, val=val, max_epochs=4, checkpoint_at_end=True, val_chunk=2048,
                         model_kwargs={"max_width": 128, "max_layers": 2, "max_batch": 256}, capacity=16)
    an = mitune.run(tr, config={"width": tune.choice([32, 64, 128]), "num_layers": tune.choice([1, 2]),
                                "learning_rate": tune.loguniform(1e-3, 1e-2), "batch_size": tune.choice([128, 256])},
                    num_samples=24, scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=4,
                                                            reduction_factor=2),
                    local_dir=str(tmp_path), verbose=0)
    assert len(an.trials) == 24 and an.stats["engine"] == "hip"
    assert an.best_result["validation_loss"] < 0.3 * float(val[1].var())
    assert an.best_checkpoint is not None


def test_concurrent_chains_match_single_stream():
    """Splitting a pack into concurrent per-stream chains changes scheduling, never a trial's arithmetic."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=4)
    cfgs = [(i, dict(CONFIGS[i % 4], learning_rate=1e-3 * (1 + i % 3))) for i in range(8)]
    out = {}
    for streams in (1, 4):
        tr = PackedTrainable(train=train, val=val, max_epochs=2, val_chunk=512,
                             model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512, "streams": streams})
        hist, states = _run(HipPackEngine, tr, cfgs, list(range(8)), 2, device="cuda", capacity=8, seed=2)
        out[streams] = (hist, states)
    for slot in range(8):
        a, b = out[1][0][-1][slot], out[4][0][-1][slot]
        assert a["validation_mse"] == pytest.approx(b["validation_mse"], rel=2e-3), slot
        assert a["train_loss"] == pytest.approx(b["train_loss"], rel=2e-3), slot


def test_layer0_splits_its_weight_gradient_further_than_wide_layers(monkeypatch):
    """12 trials of [512, 512] with weight-gradient splits allowed (``ksplits_small``; off by default): the 512x512 layer
    has enough tiles for one pass, layer 0 (20 features) does not, so the two use different split counts and the
    optimiser must add the right number of slabs per region."""
    from mitune.options import OPTIONS
    from mitune.packed import PackedTrainable, TorchPackEngine
    from mitune.packed.engine_hip import HipPackEngine
    monkeypatch.setattr(OPTIONS, "ksplits_small", 4)
    train, val = _data(n=4000, seed=6)
    tr = PackedTrainable(train=train, val=val, max_epochs=1, val_chunk=512,
                         model_kwargs={"max_width": 512, "max_layers": 2, "max_batch": 512})
    cfgs = [(i, {"hidden_sizes": [512, 512], "batch_size": 256, "learning_rate": 1e-3 * (1 + i % 3), "optimizer": "adamw",
                 "gradient_clipping": 1.0 if i % 2 else 0.0}) for i in range(12)]
    eng = HipPackEngine(tr, device="cuda", capacity=12, seed=1)
    for slot, (idx, cfg) in enumerate(cfgs):
        eng.load(slot, idx, cfg)
    eng.train_epoch()
    got = eng.evaluate()
    plan = eng._chains[0].plan
    assert (plan.ksplits, plan.ksplits0) == (1, 2)
    ref = TorchPackEngine(tr, device="cuda", capacity=12, seed=1)
    for slot, (idx, cfg) in enumerate(cfgs):
        ref.load(slot, idx, cfg)
    ref.train_epoch()
    want = ref.evaluate()
    var_y = float(val[1].var())
    for slot in range(12):
        assert got[slot]["validation_mse"] == pytest.approx(want[slot]["validation_mse"], rel=0.08, abs=0.02 * var_y), slot
        assert got[slot]["train_loss"] == pytest.approx(want[slot]["train_loss"], rel=0.08, abs=0.02 * var_y), slot


def test_bench_runs_under_the_driver_launch_line_with_two_ranks():
    """``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2``: the launch line used for N > 1, here
    with both ranks on the one GPU of the test box and a gloo metric exchange (RCCL refuses two ranks per device)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MITUNE_COMM_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--trials-per-gpu", "8", "--rows", "20000"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints the JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["config"]["trials"] == 16 and rec["value"] > 0
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True


_RCCL_SCRIPT = r"""
import os, sys, json
import torch
sys.path.insert(0, os.getcwd())
from mitune.parallel.comm import TorchDistComm
comm = TorchDistComm(backend="nccl", rank=0, world_size=1)
dev = comm.device
out = {"backend": comm.backend, "device": str(dev)}
table = torch.arange(18, dtype=torch.float64).reshape(3, 6) - 2          # CPU metric table, as the executor sends it
g = comm.all_gather(table)
out["gather_ok"] = bool(g.shape == (1, 3, 6) and g.device.type == "cpu" and torch.equal(g[0], table))
el = torch.tensor([1.25], dtype=torch.float64, device=dev)
comm.all_reduce(el, op="max")
out["max"] = float(el.item())
slab = torch.randn(1 << 20, device=dev)
ref = slab.clone()
out["bcast_ok"] = bool(torch.equal(comm.broadcast(slab, src=0), ref))
out["obj"] = comm.broadcast_object({"epoch": 3}, src=0)
grad = torch.ones(4096, device=dev, dtype=torch.bfloat16)
w = comm.all_reduce(grad, op="avg", async_op=True)
if w is not None:
    w.wait()
out["avg"] = float(grad.float().mean().item())
comm.barrier()
# the repair path of a lost rank, as far as one GPU allows: abort the RCCL group, meet through files, form the next
# group over a file store, and use it
out["delivers_errors"] = bool(comm.delivers_errors)
out["members"] = comm.shrink()
out["after"] = [comm.rank, comm.world_size]
t2 = torch.full((8,), 3.0, device=dev)
comm.all_reduce(t2, op="sum")
g2 = comm.all_gather(table)
h1, h2 = comm.all_gather_async(table), comm.all_gather_async(table + 1)          # two posted, read in order
table += 100                                                                      # the caller's tensor may change meanwhile
out["async_ok"] = bool(torch.equal(h1.wait()[0], g2[0]) and torch.equal(h2.wait()[0], g2[0] + 1))
table -= 100
out["after_ok"] = bool(float(t2.sum().item()) == 24.0 and torch.equal(g2[0], table))
out["members2"] = comm.shrink()                   # and once more: generations do not collide
out["rdzv"] = comm.rendezvous_dir
comm.barrier()
comm.close()
print("RESULT " + json.dumps(out))
"""


def test_rccl_backend_one_rank_collectives_and_group_repair(tmp_path):
    """Every collective the executors use, on the real RCCL backend (a one-rank group: the test box has one GPU), then
    ``shrink()`` on it: abort of the RCCL process group, file rendezvous, re-initialisation over a ``FileStore``, and
    collectives on the new group."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MITUNE_RENDEZVOUS_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert rec["backend"] == "nccl" and rec["device"] == "cuda:0"
    assert rec["gather_ok"] and rec["bcast_ok"] and rec["max"] == 1.25 and rec["avg"] == 1.0 and rec["obj"] == {"epoch": 3}
    assert rec["delivers_errors"] and rec["members"] == [0] and rec["members2"] == [0] and rec["after"] == [0, 1] and rec["after_ok"] and rec["async_ok"]
    assert rec["rdzv"].startswith(str(tmp_path))
    assert os.path.isdir(os.path.join(rec["rdzv"], "gen1")) and os.path.isdir(os.path.join(rec["rdzv"], "gen2"))


# ----------------------------------------------------------------------------------------- data-parallel trial on the GPU
def test_dp_engine_step_driver_tracks_the_f32_reference():
    """The C++ step driver of a data-parallel trial (own bf16 GEMMs, fused head/loss, bf16 gradient buckets, fused
    optimiser; one rank, so no collective) against the same trial trained in f32 by autograd on the CPU."""
    from mitune.parallel.dp import DPTrainer
    g = torch.Generator().manual_seed(3)
    X = torch.randn(4096, 32, generator=g)
    y = torch.tanh(X[:, 0]) + 0.5 * X[:, 1] + 0.05 * torch.randn(4096, generator=g)
    cfg = {"hidden_sizes": [256, 128, 64], "batch_size": 512, "learning_rate": 2e-3, "optimizer": "adamw", "weight_decay": 1e-3}
    ref = DPTrainer(cfg, X[:3584], y[:3584], X[3584:], y[3584:], seed=4, trial_index=2)
    gpu = DPTrainer(cfg, X[:3584].cuda(), y[:3584].cuda(), X[3584:].cuda(), y[3584:].cuda(), seed=4, trial_index=2)
    assert gpu.engine is not None and ref.engine is None
    for k, v in ref.model.state_dict().items():
        assert torch.equal(v, gpu.model.state_dict()[k])            # same initial weights
    for epoch in range(3):
        lr, lg = ref.train_epoch(), gpu.train_epoch()
        mr, mg = ref.evaluate(), gpu.evaluate()
        assert lg == pytest.approx(lr, rel=0.05, abs=0.01), (epoch, lr, lg)
        assert mg["validation_mse"] == pytest.approx(mr["validation_mse"], rel=0.08, abs=0.01), (epoch, mr, mg)
    assert gpu.step == ref.step == 21 and mg["validation_mse"] < 0.5 * float(y.var())
    # the bf16 shadow the GEMMs read is the rounded master
    assert torch.equal(gpu.model.W, gpu.model.P.to(torch.bfloat16))
    sd = gpu.model.state_dict()
    assert sd["layers.0.weight"].shape == (256, 32) and not torch.equal(sd["layers.0.weight"], ref.model.state_dict()["layers.0.weight"])


def test_dp_step_driver_with_collectives_on_the_communication_stream(monkeypatch):
    """Same trial with the per-bucket all-reduces forced on (a one-rank RCCL communicator, where they are copies): the
    driver's event / stream hand-offs between compute and communication stream must leave every number unchanged."""
    from mitune.parallel.dp import DPTrainer
    g = torch.Generator().manual_seed(5)
    X = torch.randn(3072, 64, generator=g).cuda()
    y = (torch.tanh(X[:, 0]) - 0.3 * X[:, 5]).cuda()
    cfg = {"hidden_sizes": [512, 256], "batch_size": 256, "learning_rate": 1e-3, "optimizer": "adam"}
    out = []
    from mitune.options import OPTIONS
    for force in ("0", "1"):
        monkeypatch.setattr(OPTIONS, "dp_force_collectives", force == "1")
        t = DPTrainer(cfg, X[:2560], y[:2560], X[2560:], y[2560:], seed=1, trial_index=0)
        assert (t.engine.rccl is not None) == (force == "1")
        losses = [t.train_epoch() for _ in range(2)]
        out.append((losses, t.evaluate(), t.model.P.clone(), t.reducer.bytes_reduced))
        t.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and torch.equal(out[0][2], out[1][2])
    assert out[0][3] == 0 and out[1][3] == 2 * 10 * out[1][2].numel() * 2          # bf16 bytes, every step, every element


@pytest.mark.parametrize("cfg_extra", [{"optimizer": "adamw", "weight_decay": 1e-3},
                                       {"optimizer": "sgd", "momentum": 0.9, "gradient_clipping": 0.5, "loss_function": "huber"}])
def test_dp_engine_exact_f32_form_matches_the_cpu_reference(cfg_extra, monkeypatch):
    """``dtype="f32"``: the same driver on the exact-f32 tile kernels with f32 buckets, against the autograd CPU path --
    f32 tolerances, with and without gradient clipping -- and unchanged when the per-bucket all-reduces are forced on."""
    from mitune.options import OPTIONS
    from mitune.parallel.dp import DPTrainer
    g = torch.Generator().manual_seed(3)
    X = torch.randn(4096, 32, generator=g)
    y = torch.tanh(X[:, 0]) + 0.5 * X[:, 1] + 0.05 * torch.randn(4096, generator=g)
    cfg = dict({"hidden_sizes": [256, 128, 64], "batch_size": 512, "learning_rate": 2e-3, "dtype": "f32"}, **cfg_extra)
    ref = DPTrainer(cfg, X[:3584], y[:3584], X[3584:], y[3584:], seed=4, trial_index=2)
    runs = []
    for force in (False, True):
        monkeypatch.setattr(OPTIONS, "dp_force_collectives", force)
        gpu = DPTrainer(cfg, X[:3584].cuda(), y[:3584].cuda(), X[3584:].cuda(), y[3584:].cuda(), seed=4, trial_index=2)
        assert gpu.engine is not None and gpu.engine.f32 and gpu.model.W is gpu.model.P and gpu.X.dtype == torch.float32
        assert (gpu.engine.rccl is not None) == force
        hist = []
        for epoch in range(3):
            hist.append((gpu.train_epoch(), gpu.evaluate()))
        runs.append((hist, gpu.model.P.clone(), gpu.reducer.bytes_reduced))
        gpu.close()
        assert gpu.engine.rccl is None
    for epoch in range(3):
        lr, mr = ref.train_epoch(), ref.evaluate()
        lg, mg = runs[0][0][epoch]
        assert lg == pytest.approx(lr, rel=2e-4, abs=1e-6), (epoch, lr, lg)
        for k in ("validation_loss", "validation_mse", "validation_mape"):
            assert mg[k] == pytest.approx(mr[k], rel=5e-4, abs=1e-6), (epoch, k, mr, mg)
    torch.testing.assert_close(runs[0][1].cpu(), ref.model.P, rtol=2e-3, atol=2e-5)
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][2] == 0 and runs[1][2] == 3 * 7 * runs[1][1].numel() * 4           # f32 bytes, every step, every element


def test_dp_engine_refuses_shapes_it_cannot_take_instead_of_running_eagerly():
    from mitune.parallel.dp import DPTrainer
    X, y = torch.randn(512, 30).cuda(), torch.randn(512).cuda()
    with pytest.raises(ValueError, match="multiple of 8"):
        DPTrainer({"hidden_sizes": [64], "batch_size": 64}, X, y, X[:64], y[:64])
    X = torch.randn(512, 32).cuda()
    with pytest.raises(ValueError, match="multiples of 8"):
        DPTrainer({"hidden_sizes": [60], "batch_size": 64}, X, y, X[:64], y[:64])
    with pytest.raises(ValueError, match="f32"):
        DPTrainer({"hidden_sizes": [64], "batch_size": 64, "gradient_clipping": 1.0}, X, y, X[:64], y[:64])
    with pytest.raises(ValueError, match="bf16"):
        DPTrainer({"hidden_sizes": [64], "batch_size": 64, "dtype": "bf16"}, X.cpu(), y.cpu(), X[:64].cpu(), y[:64].cpu())


def test_dp_trials_under_the_search_driver_on_the_gpu():
    """``mitune.run(dp_mlp_trainable, ...)`` with ASHA over learning rate x batch size, both compute types in one space."""
    import mitune
    from mitune import tune
    from mitune.parallel.dp import dp_mlp_trainable
    from mitune.schedulers import ASHAScheduler
    g = torch.Generator().manual_seed(9)
    X = torch.randn(4096, 32, generator=g).cuda()
    y = (torch.tanh(X[:, 0]) + 0.5 * X[:, 1]).cuda()
    tr = tune.with_parameters(dp_mlp_trainable, train=(X[:3584], y[:3584]), val=(X[3584:], y[3584:]), epochs=3, seed=1)
    an = mitune.run(tr, config={"hidden_sizes": [128, 64], "batch_size": tune.choice([256, 512]), "steps_per_epoch": 5,
                                "learning_rate": tune.loguniform(1e-3, 1e-2), "dtype": tune.choice(["bf16", "f32"])},
                    num_samples=6, max_concurrent_trials=1, seed=0, verbose=0, local_dir=None,
                    scheduler=ASHAScheduler(metric="validation_loss", mode="min", max_t=3, reduction_factor=2))
    assert len(an.trials) == 6 and {t.config["dtype"] for t in an.trials} == {"bf16", "f32"}
    assert all(np.isfinite(t.last_result["validation_loss"]) and t.last_result["steps"] == 5 * t.training_iteration
               for t in an.trials)
    assert max(t.training_iteration for t in an.trials) == 3 and min(t.training_iteration for t in an.trials) < 3


_RCCL_DIRECT = r"""
import os, sys, json
import torch
sys.path.insert(0, os.getcwd())
from mitune.parallel.comm import LocalComm
from mitune.parallel.rccl import RcclComm
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rc = RcclComm(LocalComm(dev), dev)
out = {"lib": os.path.basename(os.path.dirname(os.path.dirname(rc.lib._name))) + "/" + os.path.basename(rc.lib._name)}
t = torch.arange(4096, device=dev, dtype=torch.float32).to(torch.bfloat16)
ref = t.clone()
rc.stream.wait_stream(torch.cuda.current_stream())
rc.all_reduce(t, "avg", stream=rc.stream)
rc.all_reduce_rs_ag(t, "sum", stream=rc.stream)
torch.cuda.current_stream().wait_stream(rc.stream)
torch.cuda.synchronize()
out["same"] = bool(torch.equal(t, ref))
out["fn"] = bool(rc.fn_all_reduce) and bool(rc.comm_handle)
rc.close()
print("RESULT " + json.dumps(out))
"""


def test_direct_rccl_communicator_one_rank():
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / reduce-scatter + all-gather through parallel/rccl.py on the RCCL
    library PyTorch itself has loaded (a one-rank communicator: the box has one GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _RCCL_DIRECT], cwd=root, capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert rec["same"] and rec["fn"] and rec["lib"].endswith("librccl.so")


def test_device_record_table_matches_the_host_path():
    """The round's record table written on the device (what the ranks all-gather) carries exactly the numbers of the
    host path (train_epoch + evaluate), bit for bit, plus a time row from the GPU clock."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=9)
    for dtype in ("bf16", "f32"):
        tr = PackedTrainable(train=train, val=val, max_epochs=2, val_chunk=512, dtype=dtype,
                             model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
        a = HipPackEngine(tr, device="cuda", capacity=5, seed=3)
        b = HipPackEngine(tr, device="cuda", capacity=5, seed=3)
        for eng in (a, b):
            eng.load(1, 11, CONFIGS[0])
            eng.load(3, 12, CONFIGS[1])
        for epoch in (1, 2):
            a.train_epoch()
            host = a.evaluate()
            b.begin_round()
            b.train_epoch()
            tab = b.device_table()
            assert tab.device.type == "cuda" and tab.shape == (6, 6) and tab.dtype == torch.float64
            t = tab.cpu()
            for slot, idx in ((1, 11), (3, 12)):
                m = host[slot]
                assert t[slot].tolist() == [float(idx), float(epoch), m["train_loss"], m["validation_loss"], m["validation_mse"],
                                            m["validation_mape"]]
            for slot in (0, 2, 4):
                assert t[slot].tolist() == [-1.0] * 6
            assert t[5, 0] == -2.0 and t[5, 2] == 2.0 and 0.0 < t[5, 3] <= t[5, 1] < 5.0


def test_cross_rank_exploit_moves_device_slabs():
    """PBT exploit between two engines (= two GPUs): export_raw on the donor, adopt_raw on the receiver, no host copy of
    the weights; the receiver then continues exactly like an in-pack clone of the donor would."""
    from mitune.packed import PackedTrainable
    from mitune.packed.engine_hip import HipPackEngine
    train, val = _data(seed=10)
    tr = PackedTrainable(train=train, val=val, max_epochs=3, val_chunk=512,
                         model_kwargs={"max_width": 256, "max_layers": 3, "max_batch": 512})
    donor_cfg = dict(CONFIGS[1])
    new_cfg = dict(CONFIGS[1], learning_rate=5e-4)
    a = HipPackEngine(tr, device="cuda", capacity=4, seed=1)       # "rank 0": donor 20 and, for reference, an in-pack clone
    b = HipPackEngine(tr, device="cuda", capacity=4, seed=1)       # "rank 1": receiver 21
    a.load(0, 20, donor_cfg); a.load(1, 21, dict(CONFIGS[1], learning_rate=3e-3))
    b.load(2, 21, dict(CONFIGS[1], learning_rate=3e-3))
    a.train_epoch(); b.train_epoch()
    raw = a.export_raw(0)
    b.adopt_raw(2, raw, new_cfg)
    a.clone(0, 1, new_cfg)
    a.train_epoch(); b.train_epoch()
    ea, eb = a.export(1), b.export(2)
    for k, v in ea["model"].items():
        assert torch.equal(v, eb["model"][k]), k
    assert ea["step"] == eb["step"] and eb["trial_index"] == 21
    with pytest.raises(ValueError):
        b.adopt_raw(2, raw, dict(CONFIGS[0]))                      # another architecture
