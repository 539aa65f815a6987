set -e
mkdir -p gpurun_out/sweep
for opt in "f32_tiny_tiles=1536" "f32_tiny_tiles=2048" "f32_tiny_tiles=3072" "f32_tiny_tiles=6000" "f32_tiny_tiles=1024"; do
  name=$(echo "$opt" | tr '=:,' '___'); name=${name:-default}
  MITUNE_OPTIONS="$opt" timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/sweep/$name.json'))
print('$name', round(d['value']), round(d['elapsed_s'],3), d['best_validation_mse'])
"
done
