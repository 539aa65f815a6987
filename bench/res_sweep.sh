set -e
cd /root/repo
run() { echo "== $*" >> gpurun_out/s5_chain_sweep.txt; env "$@" timeout -k 10 200 python bench.py --no-bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['elapsed_s'], d['value'], d['best_validation_mse'])" >> gpurun_out/s5_chain_sweep.txt; }
run MITUNE_F32_CHAIN_RULE=128:4,32:2
run MITUNE_F32_CHAIN_RULE=128:4,8:2
run MITUNE_F32_CHAIN_RULE=128:4,8:2 MITUNE_CHAINS_BY_BATCH=0
run MITUNE_F32_CHAIN_RULE=128:4,8:4 MITUNE_CHAINS_BY_BATCH=0
run MITUNE_F32_CHAIN_RULE=128:4,32:2,8:3
