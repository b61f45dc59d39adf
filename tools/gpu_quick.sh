#!/bin/bash
# quick regression (parity) + bench line
mkdir -p gpurun_out
nproc > gpurun_out/nproc.txt; lscpu | grep "Model name" >> gpurun_out/nproc.txt
timeout 600 python -m pytest tests -m gpu -x -q --timeout=300 --timeout-method=thread > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests.log
timeout 600 python bench.py ${BENCH_ARGS} > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; echo "bench rc=$?" >> gpurun_out/bench_q.err
tail -5 gpurun_out/gpu_tests.log; cat gpurun_out/nproc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_q.json').read())
print('value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'roofline',d['roofline']['achieved'],d['roofline']['frac'])
print('phase',d['phase_ms_per_step']); print('kern',d['kernel_ms_per_step']); print('cpu',d.get('cpu_baseline'))
PY
