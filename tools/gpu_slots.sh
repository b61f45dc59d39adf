#!/bin/bash
mkdir -p gpurun_out
for s in 1 2 4 8; do
timeout 600 python bench.py --steps 32 --warmup 8 --cpu-rounds 0 --slots $s > gpurun_out/bench_s$s.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_s$s.json').read().strip().split('\n')[-1])
print('slots=$s value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'steps',d['steps'],'count_ms',round(d['kernel_ms_per_step']['k_count_ms'],3),'frac',round(d['roofline']['frac'],4), 'phase', {k:round(v,2) for k,v in d['phase_ms_per_step'].items()})
PY
done
