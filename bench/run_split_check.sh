set -e
mkdir -p gpurun_out/split
timeout -k 10 300 python -u -m pytest tests/test_rowblock_gpu.py tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/split/tests.log 2>&1 || { tail -30 gpurun_out/split/tests.log; exit 1; }
tail -3 gpurun_out/split/tests.log
timeout -k 10 200 python bench/small_pack_micro.py --forms 0,1s1,1s4 --slots 4 --check > gpurun_out/split/micro4e.json 2> gpurun_out/split/micro4e.err
timeout -k 10 200 python bench/small_pack_micro.py --forms 0,1s1,1s2,1s4 --check > gpurun_out/split/micro16e.json 2> gpurun_out/split/micro16e.err
python - <<'PY'
import json
for f in ('micro4e','micro16e'):
    d=json.load(open(f'gpurun_out/split/{f}.json'))
    print(f, 'n_mismatches', d['forms'].get('n_mismatches'))
    for k,v in d['forms'].items():
        if not isinstance(v,dict): continue
        print('  ',k, round(v['us_per_step'],1), v['rb_blocks'], [sorted(set(x.values())) for x in v['rb_split']], [sum(1 for n in x.values() if n>1) for x in v['rb_split']])
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-bf16 > gpurun_out/split/bench_driver.json 2> gpurun_out/split/bench_driver.err
cat gpurun_out/split/bench_driver.json
timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/split/bench_16.json 2> gpurun_out/split/bench_16.err
cat gpurun_out/split/bench_16.json
