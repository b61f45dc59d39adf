#!/bin/bash
# dense-seed regime (k=10, error-free and 3 % error): index query / chaining carry real traffic here
mkdir -p gpurun_out
for e in 0.0 0.03; do
DPH_PROFILE=1 timeout 900 python bench.py --k 10 --error $e --steps 40 --warmup 12 --cpu-rounds 0 --index-steps 20 > gpurun_out/bench_k10_$e.json 2> gpurun_out/bench_k10_$e.err; echo "k10 e=$e rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_k10_$e.json').read().strip().split('\n')[-1])
print('K10 e=$e value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'steps',d['steps'],'paf',d['paf_lines'])
print(' phase',{k:round(v,2) for k,v in d['phase_ms_per_step'].items()}); print(' kern',{k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}); print(' iq',d['index_query'])
PY
grep "thread CPU\|per executed" gpurun_out/bench_k10_$e.err | cut -c1-700
done
