"""Command line: ``python -m mitune <command>``.

    python -m mitune build [--force] [--sanitize]    compile ops/csrc/*.hip for gfx950 into ops/libmitune_hip.so
    python -m mitune run SCRIPT [args...]            run an experiment script (sets up sys.path, prints the best config)
    python -m mitune analyze EXPERIMENT_DIR [--metric M --mode min|max --top N]
    python -m mitune info                            devices, library status, HBM budget for the default pack shape

The reference has no flags or CLI at all: every path and count is a literal in its scripts
(/root/reference/ray-tune-hpo-regression.py:434-435,472-476).
"""
from __future__ import annotations

import argparse
import json
import os
import runpy
import sys


def _cmd_build(args) -> int:
    from .ops import build
    if args.sanitize:
        print(build.build_sanitized())
        return 0
    print(build.build(force=args.force))
    return 0


def _cmd_run(args) -> int:
    sys.argv = [args.script] + args.script_args
    sys.path.insert(0, os.path.dirname(os.path.abspath(args.script)))
    runpy.run_path(args.script, run_name="__main__")
    return 0


def _cmd_analyze(args) -> int:
    from .tune.analysis import ExperimentAnalysis
    an = ExperimentAnalysis.from_directory(args.experiment_dir, args.metric, args.mode)
    metric, mode = an._mm(args.metric, args.mode)
    rows = []
    for t in an.trials:
        v = t.last_result.get(metric)
        rows.append((v if isinstance(v, (int, float)) else float("inf"), t))
    rows.sort(key=lambda r: r[0], reverse=(mode == "max"))
    print(f"{len(an.trials)} trials, metric {metric} ({mode})")
    for v, t in rows[: args.top]:
        print(f"  {t.trial_id} {t.status:<10} it={t.training_iteration:<4} {metric}={v:.6g}  {json.dumps(t.config)}")
    best = an.best_trial
    if best is not None:
        print("best config:", json.dumps(best.config))
        print("best logdir:", best.logdir)
        print("best checkpoint:", an.best_checkpoint)
    return 0


def _cmd_info(args) -> int:
    import torch
    from .ops import build
    from .utils.hbm import default_capacity, mlp_pack_footprint
    print("torch", torch.__version__, "| cuda available:", torch.cuda.is_available(), "| devices:", torch.cuda.device_count())
    print("kernel library:", build.LIB, "(current)" if build.is_current() else "(stale or missing: run `python -m mitune build`)")
    fp = mlp_pack_footprint(64, 512, 3, 4096, 800_000, 200_000)
    print(f"default pack shape (64 features, width<=512, 3 layers): {fp.per_slot_bytes / 2**20:.1f} MiB per slot, "
          f"{default_capacity(64, 512, 3, 4096, 800_000, 200_000)} slots within 240 GB")
    return 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="mitune")
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build"); b.add_argument("--force", action="store_true")
    b.add_argument("--sanitize", action="store_true", help="host-side ASan/UBSan build (libmitune_hip_asan.so)")
    b.set_defaults(fn=_cmd_build)
    r = sub.add_parser("run"); r.add_argument("script"); r.add_argument("script_args", nargs=argparse.REMAINDER)
    r.set_defaults(fn=_cmd_run)
    a = sub.add_parser("analyze"); a.add_argument("experiment_dir"); a.add_argument("--metric"); a.add_argument("--mode")
    a.add_argument("--top", type=int, default=10); a.set_defaults(fn=_cmd_analyze)
    i = sub.add_parser("info"); i.set_defaults(fn=_cmd_info)
    args = ap.parse_args(argv)
    return args.fn(args)


if __name__ == "__main__":
    sys.exit(main())
