set -e
cd /root/repo
for tile in 64 32 33; do
  MITUNE_RES_PROF=1 MITUNE_RES_TILE=$tile MITUNE_RES_TIMEOUT_S=5 timeout -k 10 120 python bench/resident_micro.py --modes resident >> gpurun_out/s5_res_micro3.txt 2>&1
done
MITUNE_RES_PROF=1 MITUNE_RES_TILE=33 MITUNE_RES_WGS=768 MITUNE_RES_TIMEOUT_S=5 timeout -k 10 120 python bench/resident_micro.py --modes resident >> gpurun_out/s5_res_micro3.txt 2>&1
MITUNE_RES_PROF=1 MITUNE_RES_SYNC=2 MITUNE_RES_TILE=33 MITUNE_RES_TIMEOUT_S=5 timeout -k 10 120 python bench/resident_micro.py --modes resident >> gpurun_out/s5_res_micro3.txt 2>&1
