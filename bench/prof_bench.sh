#!/bin/bash
# Kernel trace of one bench.py search under rocprofv3, reduced on the box to two small text files (the raw trace of a
# full search is several hundred MB):   bench/prof_bench.sh <tag> [bench.py arguments ...]
#   gpurun_out/<tag>_kernel_stats.csv          rocprofv3's own per-kernel totals
#   gpurun_out/<tag>_per_epoch_summary.txt     bench/trace_summary.py over selected epochs
set -u
ROOT="$(pwd)"
TAG="$1"; shift
mkdir -p "$(dirname "$ROOT/gpurun_out/$TAG")"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$TAG"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$TAG" -- python3 "$ROOT/bench.py" "$@" \
    > "$ROOT/gpurun_out/${TAG}_bench.json" 2> "$ROOT/gpurun_out/${TAG}_bench.err" || { tail -5 "$ROOT/gpurun_out/${TAG}_bench.err"; exit 1; }
TRACE="$(find "/tmp/prof_$TAG" -name '*kernel_trace.csv' | head -1)"
STATS="$(find "/tmp/prof_$TAG" -name '*kernel_stats.csv' | head -1)"
cp "$STATS" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
python3 "$ROOT/bench/trace_summary.py" "$TRACE" 1 2 3 5 8 12 16 > "$ROOT/gpurun_out/${TAG}_per_epoch_summary.txt"
cat "$ROOT/gpurun_out/${TAG}_bench.json"
head -40 "$ROOT/gpurun_out/${TAG}_per_epoch_summary.txt"
