#!/bin/bash
for rep in 1 2; do for p in 0 1; do
DP_PLANNER_PRIORITY=$p DPH_PROFILE=1 timeout 300 python bench.py --steps 450 --cpu-rounds 0 --index-steps 0 > gpurun_out/prio.json 2> gpurun_out/prio.err
python3 - $p <<'PY'
import json,sys
d=json.loads(open('gpurun_out/prio.json').read().strip().split(chr(10))[-1]); print('priority',sys.argv[1], round(d['value']), round(d['ms_per_step'],3), d['host_cpu'])
PY
grep -o "plans computed [0-9]* ([0-9.]* ms each)[^|]*| plan wait [0-9.]* ms" gpurun_out/prio.err; grep -o "plan.speculate [0-9.]* plan.commitLoop [0-9.]*" gpurun_out/prio.err | head -1
done; done
