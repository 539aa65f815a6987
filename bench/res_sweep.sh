set -e
cd /root/repo
run() { echo "== $*" >> gpurun_out/s5_big_sweep.txt; env "$@" timeout -k 10 200 python bench.py --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['elapsed_s'], d['value'], d['best_validation_mse'])" >> gpurun_out/s5_big_sweep.txt; }
run MITUNE_BIG_MIN_TILES=0
run MITUNE_BIG_MIN_TILES=256
run MITUNE_BIG_MIN_TILES=256 MITUNE_BIG_COST=4.5
run MITUNE_BIG_MIN_TILES=256 MITUNE_BIG_COST=6
run MITUNE_BIG_MIN_TILES=64 MITUNE_BIG_COST=4.5
cat gpurun_out/s5_big_sweep.txt
