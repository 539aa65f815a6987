set -e
mkdir -p gpurun_out/full
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/full/gputests.log 2>&1 || { tail -30 gpurun_out/full/gputests.log; exit 1; }
tail -2 gpurun_out/full/gputests.log
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/full/smoke.log 2>&1 || { tail -20 gpurun_out/full/smoke.log; exit 1; }
tail -4 gpurun_out/full/smoke.log
