#!/bin/bash
# rocprofv3 kernel statistics of any python benchmark, reduced to the per-kernel totals:
#   bench/prof_cmd.sh <tag> <script.py> [arguments ...]   ->   gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>.out
set -u
ROOT="$(pwd)"
TAG="$1"; shift
SCRIPT="$ROOT/$1"; shift
mkdir -p "$(dirname "$ROOT/gpurun_out/$TAG")"
cd /tmp && export TMPDIR=/tmp
rm -rf "/tmp/prof_$TAG"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$TAG" -- python3 "$SCRIPT" "$@" \
    > "$ROOT/gpurun_out/${TAG}.out" 2> "$ROOT/gpurun_out/${TAG}.err" || { tail -5 "$ROOT/gpurun_out/${TAG}.err"; exit 1; }
STATS="$(find "/tmp/prof_$TAG" -name '*kernel_stats.csv' | head -1)"
tail -2 "$ROOT/gpurun_out/${TAG}.out" | cut -c1-600
python3 - "$STATS" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:          # same table, kernel names cut to 160 characters
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        w.writerow(dict(r, Name=r["Name"][:160]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lib = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("Cijk"))
own = sum(float(r["TotalDurationNs"]) for r in rows if "mt_" in r["Name"])
aten = sum(float(r["TotalDurationNs"]) for r in rows if "at::native" in r["Name"] or "at_cuda" in r["Name"])
print(f"kernel time {tot/1e6:.1f} ms: own mt_* {100*own/tot:.1f} %, BLAS library Cijk_* {100*lib/tot:.1f} % ({sum(1 for r in rows if r['Name'].startswith('Cijk'))} rows), aten {100*aten/tot:.1f} %")
for r in rows[:14]:
    print(f"  {float(r['Percentage']):5.1f} %  {int(r['Calls']):>7} calls  {r['Name'][:110]}")
PY
