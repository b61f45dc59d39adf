#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/bench_map.py > gpurun_out/bench_map.json 2> gpurun_out/bench_map.err; echo "map rc=$?"
timeout 900 python bench.py --k 10 --steps 3 --warmup 1 --cpu-rounds 0 > gpurun_out/bench_k10.json 2> gpurun_out/bench_k10.err; echo "k10 rc=$?"
cat gpurun_out/bench_map.json; tail -2 gpurun_out/bench_map.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_k10.json').read())
print('K10 value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'roofline',d['roofline']['achieved'])
print('phase',d['phase_ms_per_step']); print('kern',d['kernel_ms_per_step']); print('iq',d['index_query'])
PY
tail -3 gpurun_out/bench_k10.err
