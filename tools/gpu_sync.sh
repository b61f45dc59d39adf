#!/bin/bash
# how the host waits for the device: runtime blocking wait vs poll + sleep (dp_stream_sync)
mkdir -p gpurun_out/sync
run() { # name, env...
  name=$1; shift
  env "$@" DPH_PROFILE=1 timeout 300 python bench.py --cpu-rounds 0 --index-steps 0 --steps ${STEPS:-400} > gpurun_out/sync/$name.json 2> gpurun_out/sync/$name.err
  python - $name <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open('gpurun_out/sync/%s.json'%n).read().strip().split('\n')[-1])
    print(n,'value',round(d['value']),'ms/step',round(d['ms_per_step'],3),'cpu_s/step(ms)',round(1e3*d['host_cpu']['cpu_s']/d['steps'],2),'throttled',round(d['host_cpu']['throttled_s'],3))
except Exception as e:
    print(n,'ERR',e)
PY
  grep "thread CPU per round" gpurun_out/sync/$name.err
}
run block DP_SYNC_POLL_US=0
run poll20 DP_SYNC_POLL_US=20
run poll50 DP_SYNC_POLL_US=50
run poll10 DP_SYNC_POLL_US=10
run block_rocwait1 DP_SYNC_POLL_US=0 ROC_ACTIVE_WAIT_TIMEOUT=1
run poll20_b DP_SYNC_POLL_US=20
run block_b DP_SYNC_POLL_US=0
