set -e
mkdir -p gpurun_out/prio
LIB=distributed-machine-learning_amd/ops/libmitune_hip.so
for m in 0 1 2; do
  cp bench/ab/lib_mode$m.so $LIB
  for s in 16 4; do
    timeout -k 10 100 python bench/small_pack_micro.py --forms 1s1,1s4 --slots $s --check > gpurun_out/prio/m${m}_${s}.json 2> gpurun_out/prio/m${m}_${s}.err
    python -c "
import json
d=json.load(open('gpurun_out/prio/m${m}_${s}.json'))['forms']
print('prio mode $m slots $s', round(d['1s1']['us_per_step'],1), round(d['1s4']['us_per_step'],1), d.get('n_mismatches'))
"
  done
done
cp bench/ab/lib_mode0.so $LIB
