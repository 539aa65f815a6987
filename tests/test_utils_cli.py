"""Utilities, ray-shaped import paths and the command line (CPU)."""
import json
import os
import pickle
import subprocess
import sys

import pytest
import torch

import mitune
from mitune import tune

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ray_shaped_import_paths():
    from mitune.tune.schedulers import ASHAScheduler, AsyncHyperBandScheduler, PopulationBasedTraining, MedianStoppingRule, FIFOScheduler
    from mitune.tune.suggest import BasicVariantGenerator, ConcurrencyLimiter
    from mitune.tune.suggest.bayesopt import BayesOptSearch
    assert ASHAScheduler is AsyncHyperBandScheduler
    assert callable(BayesOptSearch) and callable(PopulationBasedTraining) and callable(MedianStoppingRule)
    assert callable(FIFOScheduler) and callable(BasicVariantGenerator) and callable(ConcurrencyLimiter)


def test_hbm_budget_is_monotone_and_capped():
    from mitune.utils.hbm import default_capacity, mlp_pack_footprint
    small = mlp_pack_footprint(64, 128, 3, 1024, 1000, 100)
    big = mlp_pack_footprint(64, 512, 3, 4096, 1000, 100)
    assert big.per_slot_bytes > small.per_slot_bytes > 0
    assert default_capacity(64, 512, 3, 4096, 800_000, 200_000) == 1024            # hard cap
    few = default_capacity(64, 8192, 4, 16384, 800_000, 200_000, budget_gb=40)
    assert 1 <= few < 64
    assert big.total(few) <= 40e9 or few == 1


def test_phase_timer_accumulates():
    from mitune.utils.timers import PhaseTimer
    t = PhaseTimer()
    for _ in range(3):
        with t.phase("a"):
            sum(range(1000))
    with t.phase("b"):
        pass
    s = t.summary()
    assert s["a"]["calls"] == 3 and s["b"]["calls"] == 1 and s["a"]["total_s"] >= 0


def test_profiling_commands_respect_counter_limits():
    from mitune.utils.profiling import PMC_PASSES, kernel_trace_cmd, pmc_cmd
    limits = {"SQ_": 8, "TCC_": 4, "TCP_": 4, "TA_": 2, "TD_": 2, "GRBM_": 2}
    for name, counters in PMC_PASSES.items():
        for pre, lim in limits.items():
            assert sum(c.startswith(pre) for c in counters) <= lim, name
        cmd = pmc_cmd(["bench.py", "--steps", "1"], "out", name)
        assert "--pmc" in cmd and "--sys-trace" not in cmd and "--runtime-trace" not in cmd
        assert "python" in os.path.basename(cmd[cmd.index("--") + 1])      # the interpreter itself, no wrapper
    k = kernel_trace_cmd(["bench.py"], "out")
    assert "--kernel-trace" in k and "--stats" in k and "--pmc" not in k


def test_seed_everything_reproducible():
    from mitune.utils.seeding import seed_everything
    seed_everything(7)
    a = torch.rand(3)
    seed_everything(7)
    assert torch.equal(a, torch.rand(3))


def test_params_pkl_written_next_to_params_json(tmp_path):
    def f(config):
        tune.report(loss=config["x"])
    an = tune.run(f, config={"x": tune.uniform(0, 1), "shape": (2, 3)}, num_samples=1, local_dir=str(tmp_path), verbose=0)
    d = an.trials[0].logdir
    with open(os.path.join(d, "params.pkl"), "rb") as fh:
        cfg = pickle.load(fh)        # a file this test wrote itself
    assert cfg["shape"] == (2, 3)
    assert json.load(open(os.path.join(d, "params.json")))["shape"] == [2, 3]


def test_cli_info_and_analyze(tmp_path):
    def f(config):
        for i in range(2):
            tune.report(loss=config["x"] + i)
    tune.run(f, config={"x": tune.grid_search([3.0, 1.0, 2.0])}, local_dir=str(tmp_path), name="exp", metric="loss",
             mode="min", verbose=0)
    env = dict(os.environ, PYTHONPATH=REPO)
    out = subprocess.run([sys.executable, "-m", "mitune", "analyze", str(tmp_path / "exp"), "--metric", "loss", "--mode",
                          "min", "--top", "1"], capture_output=True, text=True, env=env, cwd=REPO, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "3 trials" in out.stdout and '"x": 1.0' in out.stdout.split("best config:")[1]
    out = subprocess.run([sys.executable, "-m", "mitune", "info"], capture_output=True, text=True, env=env, cwd=REPO,
                         timeout=120)
    assert out.returncode == 0 and "kernel library" in out.stdout


def test_diverged_trial_does_not_disturb_neighbours():
    """One trial blowing up to NaN/inf must leave the other slots of the pack bit-identical."""
    from mitune.packed import PackedTrainable
    from mitune.packed.executor import PackedExecutor
    g = torch.Generator().manual_seed(0)
    X = torch.randn(512, 8, generator=g)
    y = X[:, 0] - X[:, 1]
    tr = PackedTrainable(train=(X[:384], y[:384]), val=(X[384:], y[384:]), max_epochs=3, engine="torch", dtype="f32",
                         capacity=4, model_kwargs={"max_width": 32, "max_layers": 2, "max_batch": 64})

    def search(lrs):
        ex = PackedExecutor(tr, device=torch.device("cpu"), capacity=4, seed=0)
        an = mitune.run(tr, config={"hidden_sizes": [32, 32], "batch_size": 64, "optimizer": "sgd",
                                    "learning_rate": tune.grid_search(lrs)}, executor=ex, seed=0, verbose=0,
                        local_dir=None, metric="validation_loss", mode="min")
        return {t.config["learning_rate"]: t for t in an.trials}

    clean = search([0.01, 0.02])
    mixed = search([0.01, 0.02, 1e6])
    bad = mixed[1e6].last_result["validation_loss"]
    assert not (bad == bad and abs(bad) < 1e30)          # diverged: NaN or astronomically large
    if bad != bad or bad == float("inf"):
        assert mixed[1e6].last_result.get("diverged") and mixed[1e6].training_iteration < 3     # retired at once
    for lr in (0.01, 0.02):
        assert mixed[lr].status == "TERMINATED" and mixed[lr].training_iteration == 3
        # neighbours of the exploding slot trained exactly as they do without it
        assert mixed[lr].last_result["validation_loss"] == clean[lr].last_result["validation_loss"]
    best = min((t for t in mixed.values()), key=lambda t: t.last_result["validation_loss"]
               if t.last_result["validation_loss"] == t.last_result["validation_loss"] else float("inf"))
    assert best.config["learning_rate"] in (0.01, 0.02)


def test_nonfinite_slot_can_be_an_error():
    from mitune.packed import PackedTrainable
    from mitune.packed.executor import PackedExecutor
    g = torch.Generator().manual_seed(0)
    X = torch.randn(256, 8, generator=g)
    y = X[:, 0].clone()
    y[3] = float("nan")                       # a poisoned label makes every slot's loss NaN
    tr = PackedTrainable(train=(X[:192], y[:192]), val=(X[192:], y[192:]), max_epochs=2, engine="torch", dtype="f32",
                         capacity=2, on_nonfinite="error", model_kwargs={"max_width": 16, "max_layers": 1, "max_batch": 64})
    ex = PackedExecutor(tr, device=torch.device("cpu"), capacity=2, seed=0)
    an = mitune.run(tr, config={"hidden_sizes": [16], "batch_size": 64, "learning_rate": tune.grid_search([0.01, 0.02])},
                    executor=ex, seed=0, verbose=0, local_dir=None, metric="validation_loss", mode="min",
                    raise_on_failed_trial=False)
    assert [t.status for t in an.trials] == ["ERROR", "ERROR"]
    assert "non-finite" in an.trials[0].error_msg and ex.stats["diverged"] == 2


def test_tensorboard_event_file_round_trip(tmp_path):
    import glob
    from mitune.tune.tbevents import crc32c, read_scalars

    assert crc32c(b"123456789") == 0xE3069283            # the CRC-32C check value

    def f(config):
        for i in range(3):
            tune.report(loss=config["x"] * (i + 1), acc=0.5)
    an = tune.run(f, config={"x": 2.0}, num_samples=1, local_dir=str(tmp_path), verbose=0)
    files = glob.glob(os.path.join(an.trials[0].logdir, "events.out.tfevents.*"))
    assert len(files) == 1
    sc = read_scalars(files[0])                            # verifies both checksums of every record
    losses = [(s, v) for s, tag, v in sc if tag == "ray/tune/loss"]
    assert losses == [(1, 2.0), (2, 4.0), (3, 6.0)]
    assert any(tag == "ray/tune/time_total_s" for _, tag, _ in sc)


def test_sanitizer_build_screens_launchers_and_plan_walk(tmp_path):
    """Loads ``libmitune_hip_asan.so`` (host half under AddressSanitizer + UBSan) into an equally instrumented driver that
    exercises argument validation, scratch sizing and the step driver's walk over a plan allocated exactly to size."""
    import ctypes
    import os
    import shutil
    import subprocess
    from mitune.ops import build, lib as L
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not (os.path.exists(clang) and os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no ROCm toolchain on this machine")
    asan = os.path.join(build.HERE, "libmitune_hip_asan.so")
    if build.library_digest(asan) != build._digest():
        build.build_sanitized(verbose=False)              # about a minute, once per source change
    assert build.library_digest(asan) == build._digest()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "asan_driver")
    subprocess.run([clang, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    os.path.join(root, "tests", "native", "asan_driver.cpp"), "-o", exe, "-L" + build.HERE,
                    "-lmitune_hip_asan", "-Wl,-rpath," + build.HERE, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([exe, str(L.MlpPlan.ksplits.offset), str(L.MlpPlan.ksplits0.offset)], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0 and "ASAN_DRIVER_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
