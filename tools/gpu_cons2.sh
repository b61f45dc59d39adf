#!/bin/bash
# host vs device consensus alignment with the polling stream wait (DPH_PROFILE CPU per section)
mkdir -p gpurun_out/cons
for cfg in "DP_DEVICE_CONSENSUS=0" "DP_DEVICE_CONSENSUS=1" "DP_DEVICE_CONSENSUS=0" "DP_DEVICE_CONSENSUS=1"; do
env $cfg DPH_PROFILE=1 timeout 600 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 > gpurun_out/cons/b.json 2> gpurun_out/cons/b.err
python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/cons/b.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'value',round(d['value']),'ms/step',round(d['ms_per_step'],3), 'cpu ms/step', round(1e3*d['host_cpu']['cpu_s']/d['steps'],2), 'kern', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
grep "thread CPU per round\|thread CPU up to" gpurun_out/cons/b.err
done
