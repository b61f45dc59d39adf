#!/bin/bash
mkdir -p gpurun_out/uni
timeout 900 python -m pytest tests -m gpu -x -q --timeout=300 --timeout-method=thread 2>&1 | tail -3
for m in 1 0 1 0; do
DP_NO_UNIFORM_STEP=$m DPH_PROFILE=1 timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 > gpurun_out/uni/b.json 2> gpurun_out/uni/b.err
python - $m <<'PY'
import json,sys
d=json.loads(open('gpurun_out/uni/b.json').read().strip().split('\n')[-1])
print('no_uniform',sys.argv[1],'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'cpu ms/step',round(1e3*d['host_cpu']['cpu_s']/d['steps'],2))
PY
grep "thread CPU per round" gpurun_out/uni/b.err
done
timeout 900 python tools/full_parity.py --max-rounds 120 --slots 5 --out gpurun_out/uni/parity_k13_e0.json 2>&1 | tail -1 | cut -c1-400
timeout 900 python tools/full_parity.py --max-rounds 40 --error 0.002 --slots 5 --out gpurun_out/uni/parity_k13_e0.002.json 2>&1 | tail -1 | cut -c1-400
timeout 900 python tools/full_parity.py --max-rounds 6 --k 10 --error 0.03 --slots 3 --out gpurun_out/uni/parity_k10_e0.03.json 2>&1 | tail -1 | cut -c1-400
