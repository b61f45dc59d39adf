#!/bin/bash
for rep in 1 2 3 4; do for cfg in "DP_SCAN_HOLD=1 DP_SCAN_CONCURRENCY=1 S=5" "DP_SCAN_HOLD=0 DP_SCAN_CONCURRENCY=1 S=8" "DP_SCAN_HOLD=0 DP_SCAN_CONCURRENCY=2 S=8" "DP_SCAN_HOLD=0 DP_SCAN_CONCURRENCY=1 S=6"; do
s=${cfg##*S=}
env ${cfg% S=*} timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 --slots $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$cfg rep=$rep', round(d['value']), round(d['ms_per_step'],3), round(1e3*d['host_cpu']['cpu_s']/d['steps'],2), round(d['host_cpu']['throttled_s'],3), round(d['kernel_ms_per_step']['k_count_ms'],3))"
done; done
