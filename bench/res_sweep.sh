set -e
cd /root/repo
for shape in "256 512" "256 128" "1024 64"; do
  set -- $shape
  timeout -k 10 200 python bench/gemm_micro.py --f32 --slots 13 --batch $1 --width $2 --iters 100 >> gpurun_out/s5_gemm_ku.txt 2>&1
done
