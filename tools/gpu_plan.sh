#!/bin/bash
mkdir -p gpurun_out/plan
for d in 2 1 2 1; do
DP_PLAN_WANT_DIV=$d DPH_PROFILE=1 timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 > gpurun_out/plan/b.json 2> gpurun_out/plan/b.err
python - $d <<'PY'
import json,sys
d=json.loads(open('gpurun_out/plan/b.json').read().strip().split('\n')[-1])
print('div',sys.argv[1],'value',round(d['value']),'ms/step',round(d['ms_per_step'],3))
PY
grep -o "plans computed [0-9]* ([0-9.]* ms each)" gpurun_out/plan/b.err; grep -o "plan.speculate [0-9.]* plan.commitLoop [0-9.]*" gpurun_out/plan/b.err
done
