set -e
mkdir -p gpurun_out/reg
timeout -k 10 100 python bench/small_pack_micro.py --forms 1s1,1s1g --slots 4 --stamps > gpurun_out/reg/st4.json 2> gpurun_out/reg/st4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/reg/st4.json'))['forms']
for f,v in d.items():
    print(f, round(v['us_per_step'],1))
    for s,st in v['stamps_ticks_median_by_slot'].items():
        print('  slot',s,v['widths_by_slot'][s], st)
PY
