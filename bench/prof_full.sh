#!/bin/bash
# Kernel trace of the reference's full workflow through the packed executor: bench/prof_full.sh <tag> [extra args]
set -e
trap 'rm -rf $OUT' EXIT      # the raw trace is tens of megabytes: only the summary travels back
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_full_$TAG
rm -rf $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench/full_workload.py --packed 16 --num-samples 32 --warm 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_full_$TAG.log 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" $GRAFT_REPO_ROOT/gpurun_out/prof_full_${TAG}_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows[:60]:
        w.writerow([c[:110] for c in r])
tot = sum(float(r[2]) for r in rows[1:])
aten = sum(float(r[2]) for r in rows[1:] if "at::native" in r[0] or "rocclr" in r[0])
cijk = sum(float(r[2]) for r in rows[1:] if r[0].startswith("Cijk"))
print(f"kernel time {tot/1e9:.2f} s, framework kernels {100*aten/tot:.1f} %, BLAS-library kernels {100*cijk/tot:.1f} %")
for r in rows[1:22]:
    print(f"{float(r[4]):6.2f}% calls {int(r[1]):7d} avg {float(r[3])/1e3:8.1f}us  {r[0][:100]}")
for r in rows[1:]:
    if r[0].startswith("Cijk"):
        print(f"BLAS {float(r[4]):6.2f}% calls {int(r[1]):7d} avg {float(r[3])/1e3:8.1f}us  {r[0][:150]}")
PY
grep workload $GRAFT_REPO_ROOT/gpurun_out/prof_full_$TAG.log | cut -c1-400
rm -rf $OUT      # the raw trace is tens of megabytes: only the summary travels back
