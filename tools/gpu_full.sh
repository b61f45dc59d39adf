#!/bin/bash
# whole job: every round of config 2 until the input is exhausted
mkdir -p gpurun_out
DPH_PROFILE=1 timeout 900 python bench.py --steps 1000000 --warmup 0 --cpu-rounds 0 --slots ${SLOTS:-4} > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
grep "\[pipe\]" gpurun_out/bench_full.err | head -3
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().split('\n')[-1])
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'steps',d['steps'],'paf',d['paf_lines'],'phase',{k:round(v,2) for k,v in d['phase_ms_per_step'].items()})
print('kern',{k:round(v,3) for k,v in d['kernel_ms_per_step'].items()})
PY
