set -e
mkdir -p gpurun_out/sweep2
for opt in "" "rowblock_split_cost=25000" "rowblock_split_cost=35000" "rowblock_split_cost=80000" "rowblock_wg_budget=224" "rowblock_wg_budget=288" "rowblock_split=2" "ksplits_small=2" "ksplits_small=8"; do
  name=$(echo "$opt" | tr '=:,' '___'); name=${name:-default}
  MITUNE_OPTIONS="$opt" timeout -k 10 100 python bench.py --no-bf16 --steps 20 --warmup 1 > gpurun_out/sweep2/$name.json 2> gpurun_out/sweep2/$name.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/sweep2/$name.json'))
print('$name', round(d['value']), round(d['elapsed_s'],3), d['best_validation_mse'])
"
done
