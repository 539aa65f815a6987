#!/bin/bash
# The reference's sample workload through the packed executor: grouped-launch engine against the per-slot graph engine.
# usage: bench/sample_seq_runs.sh [quick]
cd $GRAFT_REPO_ROOT
O=gpurun_out/sample_seq.jsonl
: > $O
run() { timeout -k 10 200 python -u bench/sample_workload.py "$@" 2>/dev/null | grep workload >> $O || exit 1; tail -1 $O | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['packed_slots'], 'asha' if r['asha'] else 'full', r['trials'], round(r['elapsed_s'],3), 's', round(r['trials_per_hour']), 'trials/h', r['executor'])"; }
run --packed 16 --no-asha --num-samples 64 --engine auto
run --packed 64 --no-asha --num-samples 256 --engine auto
run --packed 64 --num-samples 256 --engine auto
run --packed 128 --num-samples 2048 --engine auto
if [ "$1" != "quick" ]; then
run --packed 16 --no-asha --num-samples 64 --engine graph
run --packed 64 --num-samples 256 --engine graph
fi
