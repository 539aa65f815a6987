#!/bin/bash
# Kernel + memory-copy trace of a data-parallel trial with forced collectives:  bench/prof_dp_overlap.sh <tag> [target args]
set -u
ROOT="$(pwd)"
TAG="$1"; shift
mkdir -p "$(dirname "$ROOT/gpurun_out/$TAG")"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_dp"
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_dp -- python3 "$ROOT/bench/dp_overlap_target.py" "$@" \
    > "$ROOT/gpurun_out/${TAG}_target.txt" 2>&1 || { tail -5 "$ROOT/gpurun_out/${TAG}_target.txt"; exit 1; }
python3 "$ROOT/bench/dp_overlap_summary.py" /tmp/prof_dp > "$ROOT/gpurun_out/${TAG}_summary.txt" 2>&1
cat "$ROOT/gpurun_out/${TAG}_summary.txt"
