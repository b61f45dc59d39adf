#!/bin/bash
# device consensus alignment: parity suites, then bench with / without it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_overlap_e2e.py -m gpu -x -q --timeout=300 --timeout-method=thread 2>&1 | tail -6
for cfg in "X=1" "DP_HOST_CONSENSUS=1" "X=2"; do
env $cfg DPH_PROFILE=1 timeout 600 python bench.py --steps 300 --cpu-rounds 0 > gpurun_out/bench_cons.json 2> gpurun_out/bench_cons.err
python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/bench_cons.json').read().strip().split('\n')[-1])
print(sys.argv[1], 'value',round(d['value']),'ms/step',round(d['ms_per_step'],3), 'kern', {k:round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'cons phase', round(d['phase_ms_per_step']['t_consensus'],2))
PY
grep "thread CPU per round" gpurun_out/bench_cons.err
done
