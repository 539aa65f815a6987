#!/bin/bash
# Kernel trace of the sequence-pack micro-benchmark: bench/prof_seq.sh <slots> <tag>
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_seq_$2
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench/seq_pack_micro.py --slots $1 --engines seq --epochs 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_seq_$2.log 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" $GRAFT_REPO_ROOT/gpurun_out/prof_seq_$2_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows[:40]:
        w.writerow([c[:110] for c in r])
PY
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_seq_$2_stats.csv | cut -c1-200
