#!/bin/bash
for rep in 1 2 3; do for s in ${SLOTS_LIST:-4 5 6 7 8}; do
timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 --slots $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('slots=$s rep=$rep', round(d['value']), round(d['ms_per_step'],3), round(1e3*d['host_cpu']['cpu_s']/d['steps'],2), round(d['host_cpu']['throttled_s'],3))"
done; done
