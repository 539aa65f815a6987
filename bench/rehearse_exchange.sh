#!/bin/bash
# N ranks sharing the one GPU of the test box (gloo exchange), the node run's launch line, once per exchange mode:
#   bench/rehearse_exchange.sh <ranks> <out dir under gpurun_out> [modes ...]
set -u
N="$1"; O="gpurun_out/$2"; shift 2
MODES="${*:-lockstep rungs async}"
mkdir -p "$O"
PORT=29540
for MODE in $MODES; do
  PORT=$((PORT + 1))
  MITUNE_COMM_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
      --master-port $PORT bench.py --gpus "$N" --steps 16 --warmup 1 --no-bf16 --exchange "$MODE" \
      > "$O/bench_${N}proc_gloo_${MODE}.json" 2> "$O/bench_${N}proc_gloo_${MODE}.err" || { tail -5 "$O/bench_${N}proc_gloo_${MODE}.err"; exit 1; }
  python - "$O/bench_${N}proc_gloo_${MODE}.json" "$MODE" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d.get("extra", d)
print(sys.argv[2], "value", d.get("value"), "elapsed", e.get("elapsed_s"), "comm_s", e.get("comm_s"), "slot_epochs", e.get("slot_epochs"),
      "best", e.get("best_val_mse"), "migrations", e.get("migrations"))
PY
done
