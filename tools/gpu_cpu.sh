#!/bin/bash
mkdir -p gpurun_out
run() {
  env "$@" DPH_PROFILE=1 timeout 600 python bench.py --steps 400 --warmup 8 --cpu-rounds 0 --slots ${SLOTS:-4} > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
  python - "$@" <<PY
import json,sys
d=json.loads(open('gpurun_out/bench_c.json').read().strip().split('\n')[-1])
h=d['host_cpu']
print(' '.join(sys.argv[1:]), '| value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'| wall',round(h['wall_s'],3),'cpu_s',round(h['cpu_s'],2),'throttled_s',round(h['throttled_s'],3), '| phases', {k:round(v,2) for k,v in d['phase_ms_per_step'].items()})
PY
  grep "thread CPU per round" gpurun_out/bench_c.err
}
for cfg in "$@"; do run $cfg; done
