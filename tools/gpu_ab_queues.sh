#!/bin/bash
# hardware queues x executor slots on the default bench workload
mkdir -p gpurun_out/r03
for cfg in ${CFGS:-8:5 8:6 16:6 16:8 16:10 24:10}; do
  Q=${cfg%%:*}; S=${cfg##*:}
  GPU_MAX_HW_QUEUES=$Q timeout 600 python3 bench.py --steps ${STEPS:-5} --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots $S > gpurun_out/r03/abq_${Q}_${S}.json 2> gpurun_out/r03/abq_${Q}_${S}.err
  echo "queues=$Q slots=$S $(python3 -c "
import json
d=json.load(open('gpurun_out/r03/abq_${Q}_${S}.json')); print('%.2fM ms/job %.1f rounds_only %.4f parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['parity']['paf_sha256_matches_oracle_fixture']))")"
done
