#!/bin/bash
mkdir -p gpurun_out/rc
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_golden.py tests/test_gpu_overlap_e2e.py -m gpu -x -q --timeout=300 --timeout-method=thread 2>&1 | tail -3
for m in 1 0 1 0; do
DP_NO_ANCHORS=$m DPH_PROFILE=1 timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 > gpurun_out/rc/b.json 2> gpurun_out/rc/b.err
python - $m <<'PY'
import json,sys
d=json.loads(open('gpurun_out/rc/b.json').read().strip().split('\n')[-1])
print('no_anchors',sys.argv[1],'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'cpu ms/step',round(1e3*d['host_cpu']['cpu_s']/d['steps'],2))
PY
grep "thread CPU per round" gpurun_out/rc/b.err
done
