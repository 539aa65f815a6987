set -e
mkdir -p gpurun_out/post
timeout -k 10 300 python -u -m pytest tests/test_rowblock_gpu.py tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/post/tests.log 2>&1 || { tail -30 gpurun_out/post/tests.log; exit 1; }
tail -2 gpurun_out/post/tests.log
timeout -k 10 100 python bench.py --no-bf16 > gpurun_out/post/bench_16.json 2> gpurun_out/post/bench_16.err
cut -c1-200 gpurun_out/post/bench_16.json
