#!/usr/bin/env python
"""bench.py under cProfile, one report per rank: where the host time of a multi-rank run goes.

    MITUNE_COMM_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29544 bench/rehearse_profile.py --gpus 2 --steps 16 --trials-per-gpu 64 --rows 200000
"""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench as headline  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    prof = cProfile.Profile()
    prof.enable()
    try:
        headline.main()
    finally:
        prof.disable()
        out = io.StringIO()
        pstats.Stats(prof, stream=out).sort_stats("cumulative").print_stats(45)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"rehearse_profile_rank{rank}.txt"), "w") as f:
            f.write(out.getvalue())


if __name__ == "__main__":
    main()
