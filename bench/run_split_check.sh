set -e
mkdir -p gpurun_out/ob
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_rowblock_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/ob/tests.log 2>&1 || { tail -30 gpurun_out/ob/tests.log; exit 1; }
tail -2 gpurun_out/ob/tests.log
timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/ob/bench_16.json 2> gpurun_out/ob/bench_16.err
cat gpurun_out/ob/bench_16.json
timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/ob/bench_16b.json 2> gpurun_out/ob/bench_16b.err
cat gpurun_out/ob/bench_16b.json
