#!/bin/bash
for rep in 1 2 3; do for t in ${THREADS_LIST:-13 11 10 9 8}; do
DP_HOST_THREADS=$t timeout 300 python bench.py --steps 400 --cpu-rounds 0 --index-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('threads=$t rep=$rep', round(d['value']), round(d['ms_per_step'],3), d['host_cpu'])"
done; done
