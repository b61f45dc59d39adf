#!/bin/bash
mkdir -p gpurun_out
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null); nproc $(nproc)"
for s in ${SLOTS:-1 4 4 8}; do
DPH_PROFILE=1 timeout 600 python bench.py --steps ${STEPS:-64} --warmup 8 --cpu-rounds 0 --slots $s > gpurun_out/bench_p$s.json 2> gpurun_out/bench_p$s.err
grep "\[pipe\]" gpurun_out/bench_p$s.err | tail -2
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_p$s.json').read().strip().split('\n')[-1])
print('slots=$s value',round(d['value']),'ms/step',round(d['ms_per_step'],2),'steps',d['steps'], 'phase', {k:round(v,2) for k,v in d['phase_ms_per_step'].items()})
PY
done
