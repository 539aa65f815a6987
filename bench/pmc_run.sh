#!/bin/bash
# Hardware-counter passes over bench/pmc_target.py, one rocprofv3 run per pass (counters never share a run with
# system/runtime tracing; each pass stays inside the per-block counter limits).  Run from the repo root on a GPU box.
# PMC_TARGET=bench/pmc_target_seq.py PMC_TAG=seq selects the sequence-model workload and a separate output directory;
# PMC_TARGET=bench/pmc_target_f32.py PMC_TAG=f32 PMC_MFMA_OPS=SQ_INSTS_VALU_MFMA_MOPS_F32 the exact-f32 kernels.
set -u
ROOT="$(pwd)"
TARGET="$ROOT/${PMC_TARGET:-bench/pmc_target.py}"
OUT="$ROOT/gpurun_out/pmc${PMC_TAG:+_$PMC_TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 -c "import torch" || exit 1
run_pass() {
  local name="$1"; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "/tmp/pmc_$name" -- python3 "$TARGET" 3 \
      > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  python3 "$ROOT/bench/pmc_summary.py" "/tmp/pmc_$name" > "$OUT/pmc_$name.txt" || return 1
  echo "pass $name ok"
}
run_pass mfma SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ${PMC_MFMA_OPS:-SQ_INSTS_VALU_MFMA_MOPS_BF16} SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
 && run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM \
 && run_pass hbm_read FETCH_SIZE TCC_HIT_sum \
 && run_pass hbm_write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
