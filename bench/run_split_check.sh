set -e
mkdir -p gpurun_out/rep
timeout -k 10 300 python -u -m pytest tests/test_rowblock_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/rep/tests.log 2>&1 || { tail -30 gpurun_out/rep/tests.log; exit 1; }
tail -2 gpurun_out/rep/tests.log
timeout -k 10 200 python bench/small_pack_micro.py --forms 1s4r1,1s4r2,1s4r3,1s4r4 --check > gpurun_out/rep/micro16.json 2> gpurun_out/rep/micro16.err
timeout -k 10 200 python bench/small_pack_micro.py --forms 1s4r1,1s4r4 --slots 4 --check > gpurun_out/rep/micro4.json 2> gpurun_out/rep/micro4.err
python - <<'PY'
import json
for f in ('micro16','micro4'):
    d=json.load(open(f'gpurun_out/rep/{f}.json'))
    print(f, 'n_mismatches', d['forms'].get('n_mismatches'))
    for k,v in d['forms'].items():
        if not isinstance(v,dict): continue
        print('  ',k, round(v['us_per_step'],1), v['rb_blocks'], v['rb_modes'])
PY
