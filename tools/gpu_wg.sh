#!/bin/bash
for rep in 1 2 3; do for w in 2 1; do for s in 6 10; do
DP_SCAN_WG_PER_CU=$w timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 --slots $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('wg_per_cu=$w slots=$s rep=$rep', round(d['value']), round(d['ms_per_step'],3), round(1e3*d['host_cpu']['cpu_s']/d['steps'],2), round(d['host_cpu']['throttled_s'],3), {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if k in ('k_count_ms','k_write_ms','k_chain_ms')})"
done; done; done
