#!/bin/bash
mkdir -p gpurun_out/aff
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
}
B="python bench.py --steps 400 --cpu-rounds 0 --index-steps 0"
run n1_32 taskset -c 64-95 $B
run n1_24 taskset -c 64-87 $B
run n1_48 taskset -c 64-111 $B
run n1_64 taskset -c 64-127 $B
run n1_64smt taskset -c 64-127,192-255 $B
run n0_32 taskset -c 0-31 $B
run n1_32b taskset -c 64-95 $B
run n1_20 taskset -c 64-83 $B
run free $B
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head; ls /sys/class/drm/ | head; cat /sys/class/drm/card*/device/local_cpulist 2>/dev/null | head -3
python - <<'PY'
import ctypes
h=ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
buf=ctypes.create_string_buffer(64)
print(h.hipDeviceGetPCIBusId(buf,64,0), buf.value)
import os
p="/sys/bus/pci/devices/%s/"%buf.value.decode().lower()
for f in ("numa_node","local_cpulist"):
    try: print(f, open(p+f).read().strip())
    except Exception as e: print(f,"ERR",e)
PY
