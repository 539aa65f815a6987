#!/bin/bash
# Regression pass at HEAD on a one-GPU box: the whole GPU suite, the headline, the reference's two workflows, the node
# configurations that fit one GPU, and a four-rank rehearsal of the bench launch line (gloo, ranks sharing the GPU).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench.err && cut -c1-160 $O/bench.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench_driver_line.err && cut -c1-160 $O/bench_driver_line.json
timeout -k 10 300 python bench/full_workload.py --packed 16 --max-layers 16 2>/dev/null | grep -i workload > $O/full_workload.json; cut -c1-300 $O/full_workload.json
timeout -k 10 200 python bench/sample_workload.py --packed 64 --no-asha --num-samples 256 2>/dev/null | grep workload > $O/sample_workload.json; cut -c1-200 $O/sample_workload.json
timeout -k 10 300 python bench/run_config.py 4 2>/dev/null | tail -1 > $O/config4.json; cut -c1-300 $O/config4.json
timeout -k 10 300 python bench/run_config.py 5 --steps5 4 2>/dev/null | tail -1 > $O/config5.json; cut -c1-300 $O/config5.json
MITUNE_COMM_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 16 --warmup 1 --no-bf16 > $O/bench_4proc_gloo.json 2> $O/bench_4proc_gloo.err; cut -c1-200 $O/bench_4proc_gloo.json
