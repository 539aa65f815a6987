import json
import math
import os

import numpy as np
import time

import pytest
from hypothesis import given, settings, strategies as st

import mitune
from mitune import tune
from mitune.schedulers import ASHAScheduler, FIFOScheduler, MedianStoppingRule, PopulationBasedTraining, explore
from mitune.schedulers.base import CONTINUE, STOP
from mitune.search import BasicVariantGenerator, BayesOptSearch, ConcurrencyLimiter, GaussianProcess, Searcher
from mitune.tune import space as sp
from mitune.tune.trial import Trial


# ----------------------------------------------------------------------------------------- search space
def test_domains_sample_inside_their_support():
    rng = np.random.default_rng(0)
    for _ in range(200):
        assert tune.choice([1, 2, 3]).sample(rng) in (1, 2, 3)
        assert 0.1 <= tune.uniform(0.1, 0.5).sample(rng) < 0.5
        v = tune.loguniform(1e-4, 1e-2).sample(rng)
        assert 1e-4 <= v <= 1e-2
        assert tune.randint(3, 7).sample(rng) in (3, 4, 5, 6)
        q = tune.quniform(0, 1, 0.25).sample(rng)
        assert q in (0.0, 0.25, 0.5, 0.75, 1.0)


def test_loguniform_is_uniform_in_log_space():
    rng = np.random.default_rng(1)
    xs = np.log10([tune.loguniform(1e-5, 1e-1).sample(rng) for _ in range(4000)])
    hist, _ = np.histogram(xs, bins=4, range=(-5, -1))
    assert hist.min() > 850 and hist.max() < 1150


def test_sample_from_dependency_order_and_domain_return():
    space = {
        "dim_feedforward": tune.sample_from(lambda spec: spec.config.d_model * tune.choice([2, 3, 4])),
        "twice": tune.sample_from(lambda spec: spec.config["dim_feedforward"] * 2),
        "d_model": tune.choice([64, 128]),
        "nested": {"k": tune.sample_from(lambda spec: spec.config.d_model + 1)},
    }
    for cfg in sp.generate_variants(space, 20, seed=3):
        assert cfg["dim_feedforward"] in {cfg["d_model"] * m for m in (2, 3, 4)}
        assert cfg["twice"] == 2 * cfg["dim_feedforward"]
        assert cfg["nested"]["k"] == cfg["d_model"] + 1


def test_circular_sample_from_is_reported():
    space = {"a": tune.sample_from(lambda s: s.config.b), "b": tune.sample_from(lambda s: s.config.a)}
    with pytest.raises(ValueError, match="circular"):
        sp.resolve(space, np.random.default_rng(0))


def test_variants_are_reproducible_and_grid_multiplies():
    space = {"x": tune.uniform(0, 1), "g": tune.grid_search([1, 2, 3]), "c": 7}
    a = list(sp.generate_variants(space, 2, seed=5))
    b = list(sp.generate_variants(space, 2, seed=5))
    assert a == b and len(a) == 6 and sp.count_variants(space, 2) == 6
    assert sorted(v["g"] for v in a) == [1, 1, 2, 2, 3, 3] and all(v["c"] == 7 for v in a)


@given(st.integers(2, 2000), st.integers(0, 2 ** 31 - 1), st.integers(0, 1000), st.integers(0, 50))
@settings(max_examples=40, deadline=None)
def test_feistel_permutation_is_a_bijection(n, seed, trial, epoch):
    from mitune.packed.permutation import epoch_permutation
    p = epoch_permutation(n, seed, trial, epoch)
    assert sorted(p.tolist()) == list(range(n))


def test_permutations_differ_across_trials_and_epochs():
    from mitune.packed.permutation import epoch_permutation
    a, b, c = epoch_permutation(1000, 1, 2, 3), epoch_permutation(1000, 1, 2, 4), epoch_permutation(1000, 1, 3, 3)
    assert (a != b).mean() > 0.9 and (a != c).mean() > 0.9


# ----------------------------------------------------------------------------------------- ASHA
class _FakeRunner:
    def __init__(self, trials=()):
        self.trials = list(trials)


def _res(t, v, metric="loss"):
    return {"training_iteration": t, metric: v}


def test_asha_rungs_follow_grace_times_rf_powers():
    s = ASHAScheduler(metric="loss", mode="min")
    assert s.milestones == [1, 4, 16, 64]
    s = ASHAScheduler(metric="loss", mode="min", max_t=16, grace_period=2, reduction_factor=2)
    assert s.milestones == [2, 4, 8, 16]


def test_asha_async_cutoff_matches_hand_computation():
    s = ASHAScheduler(metric="loss", mode="min", max_t=100, grace_period=1, reduction_factor=4)
    r = _FakeRunner()
    trials = [Trial({}, trial_id=f"t{i}") for i in range(6)]
    losses = [0.5, 0.4, 0.9, 0.3, 0.8, 0.35]
    decisions = [s.on_trial_result(r, t, _res(1, l)) for t, l in zip(trials, losses)]
    # recorded (negated): cutoff = 75th percentile of what was seen BEFORE each arrival
    expect = []
    seen = []
    for l in losses:
        v = -l
        if seen and v < np.nanpercentile(seen, 75):
            expect.append(STOP)
        else:
            expect.append(CONTINUE)
        seen.append(v)
    assert decisions == expect
    assert decisions[0] == CONTINUE and decisions[2] == STOP and decisions[3] == CONTINUE


def test_asha_stops_at_max_t_and_on_nan():
    s = ASHAScheduler(metric="loss", mode="min", max_t=4)
    r = _FakeRunner()
    t = Trial({}, trial_id="a")
    assert s.on_trial_result(r, t, _res(1, 0.1)) == CONTINUE
    assert s.on_trial_result(r, t, _res(4, 0.1)) == STOP
    assert s.on_trial_result(r, Trial({}, trial_id="b"), _res(1, float("nan"))) == STOP


def test_asha_batch_form_is_successive_halving_and_order_independent():
    losses = {f"t{i:02d}": 1.0 - 0.05 * i for i in range(16)}
    outs = []
    for order in (sorted(losses), sorted(losses, reverse=True)):
        s = ASHAScheduler(metric="loss", mode="min", max_t=16, reduction_factor=4)
        trials = [Trial({}, trial_id=k) for k in order]
        d = s.on_batch_result(_FakeRunner(), trials, [_res(1, losses[k]) for k in order])
        outs.append({k: a for k, a in zip(order, d)})
    assert outs[0] == outs[1]
    kept = [k for k, a in outs[0].items() if a == CONTINUE]
    assert sorted(kept) == ["t12", "t13", "t14", "t15"]           # the best quarter survives rung 1


def test_asha_state_roundtrip():
    s = ASHAScheduler(metric="loss", mode="min")
    r = _FakeRunner()
    for i in range(5):
        s.on_trial_result(r, Trial({}, trial_id=f"t{i}"), _res(1, 0.1 * i))
    s2 = ASHAScheduler(metric="loss", mode="min")
    s2.load_state_dict(json.loads(json.dumps(s.state_dict())))
    assert s2.on_trial_result(r, Trial({}, trial_id="new"), _res(1, 0.39)) == s.on_trial_result(
        r, Trial({}, trial_id="new"), _res(1, 0.39))


def test_median_stopping_rule():
    s = MedianStoppingRule(metric="loss", mode="min", grace_period=2, min_samples_required=2)
    r = _FakeRunner()
    good = [Trial({}, trial_id=f"g{i}") for i in range(3)]
    for t in (1, 2, 3):
        for g in good:
            assert s.on_trial_result(r, g, _res(t, 0.1)) == CONTINUE
    bad = Trial({}, trial_id="bad")
    assert s.on_trial_result(r, bad, _res(1, 5.0)) == CONTINUE          # inside the grace period
    assert s.on_trial_result(r, bad, _res(2, 5.0)) == STOP


# ----------------------------------------------------------------------------------------- PBT
def test_explore_perturbs_resamples_and_clips():
    rng = np.random.default_rng(0)
    mut = {"lr": tune.loguniform(1e-4, 1e-2), "bs": [32, 64, 128], "wd": tune.uniform(0.0, 0.1)}
    cfg = {"lr": 1e-3, "bs": 64, "wd": 0.09, "fixed": 1}
    for _ in range(100):
        new = explore(cfg, mut, 0.25, rng)
        assert new["fixed"] == 1 and new["bs"] in (32, 64, 128)
        assert 1e-4 <= new["lr"] <= 1e-2 and 0.0 <= new["wd"] <= 0.1
    new = explore(cfg, mut, 0.0, np.random.default_rng(1))
    assert new["lr"] in (pytest.approx(1.2e-3), pytest.approx(0.8e-3))


def test_pbt_bottom_quantile_copies_top_quantile():
    s = PopulationBasedTraining(metric="loss", mode="min", perturbation_interval=2,
                                hyperparam_mutations={"lr": tune.loguniform(1e-4, 1e-1)}, quantile_fraction=0.25, seed=0)
    trials = [Trial({"lr": 10.0 ** -(i % 4 + 1)}, trial_id=f"t{i}") for i in range(8)]
    runner = _FakeRunner(trials)
    runner.executor_moves_weights = True
    losses = [0.9, 0.1, 0.5, 0.6, 0.7, 0.2, 0.8, 0.4]
    d1 = s.on_batch_result(runner, trials, [_res(1, l) for l in losses])
    assert set(d1) == {CONTINUE}                                       # interval not reached
    d2 = s.on_batch_result(runner, trials, [_res(2, l) for l in losses])
    restarted = [t.trial_id for t, d in zip(trials, d2) if d == s.RESTART]
    assert sorted(restarted) == ["t0", "t6"]                            # the two worst
    for t in trials:
        if t.trial_id in restarted:
            assert t.pbt_request["donor"] in ("t1", "t5")               # the two best
            assert 1e-4 <= t.pbt_request["config"]["lr"] <= 1e-1


# ----------------------------------------------------------------------------------------- searchers
def test_gp_interpolates_and_is_uncertain_far_away():
    x = np.linspace(0, 1, 9)[:, None]
    y = np.sin(6 * x[:, 0])
    gp = GaussianProcess().fit(x, y)
    mu, sd = gp.predict(x)
    assert np.allclose(mu, y, atol=1e-3) and sd.max() < 1e-2
    _, sd_far = gp.predict(np.array([[3.0]]))
    assert sd_far[0] > 0.5 * y.std()


def test_bayesopt_beats_random_on_a_smooth_function_with_categoricals():
    space = {"x": tune.uniform(-2, 2), "y": tune.loguniform(1e-3, 1e1), "kind": tune.choice(["a", "b", "c"]),
             "width": tune.choice([64, 128, 256, 512]),
             "derived": tune.sample_from(lambda spec: spec.config.width * 2)}

    def f(cfg):
        pen = {"a": 0.5, "b": 0.0, "c": 1.0}[cfg["kind"]]
        return (cfg["x"] - 0.7) ** 2 + (math.log10(cfg["y"]) + 1) ** 2 + pen + abs(math.log2(cfg["width"]) - 8) * 0.2

    def run(searcher, n):
        best = math.inf
        for i in range(n):
            cfg = searcher.suggest(str(i))
            assert cfg["derived"] == 2 * cfg["width"]
            v = f(cfg)
            best = min(best, v)
            searcher.on_trial_complete(str(i), {"loss": v})
        return best

    bo = BayesOptSearch(space, metric="loss", mode="min", random_search_steps=8, seed=0, num_candidates=512)
    rnd = BasicVariantGenerator(num_samples=40, seed=0)
    rnd.set_search_properties("loss", "min", space)
    b_bo, b_rnd = run(bo, 40), run(rnd, 40)
    assert b_bo < 0.15 and b_bo <= b_rnd


def test_bayesopt_state_roundtrip_and_limiter():
    space = {"x": tune.uniform(0, 1)}
    s = BayesOptSearch(space, metric="m", mode="max", random_search_steps=2, seed=1)
    for i in range(4):
        c = s.suggest(str(i))
        s.on_trial_complete(str(i), {"m": -(c["x"] - 0.5) ** 2})
    s2 = BayesOptSearch(space, metric="m", mode="max", random_search_steps=2, seed=9)
    s2.load_state_dict(json.loads(json.dumps(s.state_dict())))
    assert s2.suggest("n")["x"] == s.suggest("n")["x"]
    lim = ConcurrencyLimiter(BayesOptSearch(space, metric="m", mode="max"), max_concurrent=2)
    assert isinstance(lim.suggest("a"), dict) and isinstance(lim.suggest("b"), dict) and lim.suggest("c") is None
    lim.on_trial_complete("a", {"m": 1.0})
    assert isinstance(lim.suggest("c"), dict)


# ----------------------------------------------------------------------------------------- run(): layout, analysis, faults
def _quad(config):
    for i in range(4):
        tune.report(validation_loss=(config["x"] - 0.25) ** 2 + 1.0 / (i + 1), validation_mape=10.0 * config["x"])


def test_result_tree_layout_and_analysis(tmp_path):
    an = mitune.run(_quad, config={"x": tune.uniform(0, 1)}, num_samples=5, metric="validation_loss", mode="min",
                    local_dir=str(tmp_path), name="exp", verbose=0)
    exp = tmp_path / "exp"
    state = json.loads((exp / "experiment_state.json").read_text())
    assert len(state["trials"]) == 5 and state["metric"] == "validation_loss"
    for t in an.trials:
        d = t.logdir
        assert os.path.dirname(d) == str(exp) and os.path.basename(d).startswith("_quad_")
        assert json.loads(open(os.path.join(d, "params.json")).read()) == t.config
        lines = [json.loads(l) for l in open(os.path.join(d, "result.json"))]
        assert [l["training_iteration"] for l in lines] == [1, 2, 3, 4]
        for k in ("time_this_iter_s", "time_total_s", "timestamp", "date", "done", "trial_id", "experiment_id", "pid",
                  "hostname", "node_ip", "iterations_since_restore", "config"):
            assert k in lines[-1]
        assert open(os.path.join(d, "progress.csv")).readline().count(",") > 5
    best = an.best_trial
    assert best.config == an.best_config == an.get_best_config("validation_loss", "min")
    assert abs(best.config["x"] - 0.25) == min(abs(t.config["x"] - 0.25) for t in an.trials)
    assert an.get_best_config("validation_mape", "max")["x"] == max(t.config["x"] for t in an.trials)
    df = an.results_df
    assert len(df) == 5 and "config/x" in df.columns and "validation_loss" in df.columns
    assert set(an.trial_dataframes) == {t.trial_id for t in an.trials}
    again = mitune.ExperimentAnalysis.from_directory(str(exp))
    assert again.best_config == an.best_config


def test_metric_and_mode_are_inherited_from_the_scheduler():
    an = mitune.run(_quad, config={"x": tune.uniform(0, 1)}, num_samples=3,
                    scheduler=ASHAScheduler(metric="validation_loss", mode="min"), verbose=0)
    assert an.best_config is not None                       # would need metric/mode on run() otherwise
    an2 = mitune.run(_quad, config={"x": tune.uniform(0, 1)}, num_samples=2, verbose=0)
    with pytest.raises(ValueError, match="metric"):
        an2.best_config


def test_failed_trial_is_isolated_and_reported(tmp_path):
    def flaky(config):
        tune.report(validation_loss=config["x"])
        if config["x"] > 0.5:
            raise RuntimeError("boom at x=%.2f" % config["x"])
        tune.report(validation_loss=config["x"] / 2)

    with pytest.raises(mitune.TuneError) as ei:
        mitune.run(flaky, config={"x": tune.grid_search([0.1, 0.9, 0.2])}, metric="validation_loss", mode="min",
                   local_dir=str(tmp_path), verbose=0)
    an = ei.value.analysis
    status = {t.config["x"]: t.status for t in an.trials}
    assert status == {0.1: "TERMINATED", 0.9: "ERROR", 0.2: "TERMINATED"}
    bad = next(t for t in an.trials if t.status == "ERROR")
    assert "boom" in open(os.path.join(bad.logdir, "error.txt")).read()
    an2 = mitune.run(flaky, config={"x": tune.grid_search([0.1, 0.9])}, metric="validation_loss", mode="min",
                     raise_on_failed_trial=False, verbose=0)
    assert an2.best_config["x"] == 0.1


def test_max_failures_retries_from_the_last_checkpoint():
    attempts = {"n": 0}

    def fragile(config):
        ck = tune.get_checkpoint()
        start = ck["it"] if ck else 0
        for i in range(start, 4):
            if i == 2 and attempts["n"] == 0:
                attempts["n"] += 1
                raise RuntimeError("transient")
            tune.report(validation_loss=1.0 / (i + 1), checkpoint={"it": i + 1})

    an = mitune.run(fragile, config={}, num_samples=1, metric="validation_loss", mode="min", max_failures=1, verbose=0)
    t = an.trials[0]
    assert t.status == "TERMINATED" and t.num_failures == 1
    assert [r["training_iteration"] for r in t.results] == [1, 2, 3, 4]


def test_stop_criterion_and_time_attrs():
    an = mitune.run(_quad, config={"x": 0.5}, num_samples=1, stop={"training_iteration": 2}, metric="validation_loss",
                    mode="min", verbose=0)
    assert an.trials[0].training_iteration == 2


def test_resume_continues_unfinished_experiment(tmp_path):
    calls = []

    def slow(config):
        ck = tune.get_checkpoint()
        start = ck["it"] if ck else 0
        calls.append((config["x"], start))
        for i in range(start, 3):
            tune.report(validation_loss=config["x"] + 1.0 / (i + 1), checkpoint={"it": i + 1})

    kw = dict(config={"x": tune.grid_search([0.1, 0.2, 0.3])}, metric="validation_loss", mode="min",
              local_dir=str(tmp_path), name="r", verbose=0)
    mitune.run(slow, **kw)
    # simulate a driver that died while trial 2 was mid-flight after its first checkpoint
    path = tmp_path / "r" / "experiment_state.json"
    state = json.loads(path.read_text())
    state["trials"][2]["status"] = "RUNNING"
    path.write_text(json.dumps(state))
    import shutil
    tdir = state["trials"][2]["logdir"]
    for name in ("checkpoint_000002", "checkpoint_000003"):
        shutil.rmtree(os.path.join(tdir, name))
    calls.clear()
    an = mitune.run(slow, resume=True, **kw)
    assert calls == [(0.3, 1)]                      # only the unfinished trial ran, from its checkpoint
    assert len(an.trials) == 3 and all(t.status == "TERMINATED" for t in an.trials)
    assert an.trials[2].training_iteration == 3


def test_process_executor_runs_trials_in_worker_processes():
    an = mitune.run(_quad, config={"x": tune.uniform(0, 1)}, num_samples=2, metric="validation_loss", mode="min",
                    executor="process", max_concurrent_trials=2, verbose=0)
    pids = {t.last_result["pid"] for t in an.trials}
    assert os.getpid() not in pids and len(an.trials) == 2


def test_pbt_with_function_trainables_restarts_from_donor_checkpoint():
    seen = []

    def trainable(config):
        ck = tune.get_checkpoint()
        w = ck["w"] if ck else 0.0
        start = ck["it"] if ck else 0
        seen.append((round(config["lr"], 6), start))
        for i in range(start, 8):
            w += config["lr"]
            time.sleep(0.02)                                 # both trial threads advance side by side on a loaded box
            tune.report(score=w, checkpoint={"w": w, "it": i + 1})

    pbt = PopulationBasedTraining(metric="score", mode="max", perturbation_interval=2,
                                  hyperparam_mutations={"lr": tune.uniform(0.1, 1.0)}, quantile_fraction=0.5, seed=0)
    an = mitune.run(trainable, config={"lr": tune.grid_search([0.1, 1.0])}, scheduler=pbt, verbose=0,
                    max_concurrent_trials=2)
    assert an.stats["num_restarts"] >= 1
    assert any(start > 0 for _, start in seen)               # someone resumed from a donor checkpoint
    assert min(t.last_result["score"] for t in an.trials) > 0.1 * 8   # the weak trial was lifted by exploiting


def test_bayesopt_ignores_non_finite_observations():
    from mitune import tune
    from mitune.search.bayesopt import BayesOptSearch
    s = BayesOptSearch(metric="loss", mode="min", random_search_steps=3, seed=0)
    s.set_search_properties("loss", "min", {"x": tune.uniform(0.0, 1.0)})
    for i, v in enumerate([0.5, float("inf"), float("-inf"), float("nan"), 0.2, 0.9, 0.4]):
        cfg = s.suggest(f"t{i}")
        s.on_trial_complete(f"t{i}", {"loss": v})
    assert len(s._y) == 4
    for i in range(7, 12):                       # GP phase: every suggestion is a finite point of the space
        cfg = s.suggest(f"t{i}")
        assert 0.0 <= cfg["x"] <= 1.0
        s.on_trial_complete(f"t{i}", {"loss": (cfg["x"] - 0.3) ** 2})


def test_process_executor_maps_devices_through_the_parents_mask():
    from mitune.tune.executor import physical_device_ids
    assert physical_device_ids([0, 1], env={}) == ["0", "1"]
    assert physical_device_ids([0, 1], env={"HIP_VISIBLE_DEVICES": "4,5"}) == ["4", "5"]
    assert physical_device_ids([1], env={"CUDA_VISIBLE_DEVICES": "2, 7"}) == ["7"]
    assert physical_device_ids([1], env={"HIP_VISIBLE_DEVICES": "GPU-aa,GPU-bb", "CUDA_VISIBLE_DEVICES": "0,1"}) == ["GPU-bb"]
    import pytest
    with pytest.raises(ValueError):
        physical_device_ids([2], env={"HIP_VISIBLE_DEVICES": "4,5"})
