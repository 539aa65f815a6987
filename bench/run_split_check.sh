set -e
mkdir -p gpurun_out/ex
timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/ex/lock.json 2> gpurun_out/ex/lock.err
MITUNE_BENCH_EXCHANGE_1GPU=1 timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/ex/rungs.json 2> gpurun_out/ex/rungs.err || tail -5 gpurun_out/ex/rungs.err
MITUNE_BENCH_EXCHANGE_1GPU=1 timeout -k 10 100 python bench.py --no-bf16 --steps 20 --warmup 5 > gpurun_out/ex/rungs20.json 2> gpurun_out/ex/rungs20.err || tail -5 gpurun_out/ex/rungs20.err
timeout -k 10 100 python bench.py --no-bf16 --steps 20 --warmup 5 > gpurun_out/ex/lock20.json 2> gpurun_out/ex/lock20.err
python - <<'PY'
import json
for n in ('lock','rungs','lock20','rungs20'):
    d=json.load(open(f'gpurun_out/ex/{n}.json'))
    print(n, round(d['value']), round(d['elapsed_s'],3), d['best_validation_mse'], d['slot_epochs'], d['train_s'], d['eval_s'])
PY
