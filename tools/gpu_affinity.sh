#!/bin/bash
# does keeping the host threads on few CCDs (shared L3) help the consensus items? (they read what the slot threads wrote)
mkdir -p gpurun_out/aff
lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|L3" | head -12
cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list
nvidia-smi >/dev/null 2>&1; rocm-smi --showtoponuma 2>/dev/null | head -8
run() {
  name=$1; shift
  "$@" > gpurun_out/aff/$name.json 2> gpurun_out/aff/$name.err
  python - $name <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/aff/%s.json'%n).read().strip().split('\n')[-1])
    print(n,'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'cpu ms/step',round(1e3*d['host_cpu']['cpu_s']/d['steps'],2),'throttled',round(d['host_cpu']['throttled_s'],3))
except Exception as e:
    print(n,'ERR',e)
PY
  grep "thread CPU per round" gpurun_out/aff/$name.err
}
B="python bench.py --steps 400 --cpu-rounds 0 --index-steps 0"
export DPH_PROFILE=1
run free $B
run c0_15 taskset -c 0-15 $B
run c0_15_smt taskset -c 0-15,128-143 $B
run c0_7_smt taskset -c 0-7,128-135 $B
run c0_31 taskset -c 0-31 $B
run c64_79 taskset -c 64-79 $B
run free2 $B
