#!/bin/bash
for rep in 1 2; do for s in 3 4 5 6; do
timeout 300 python bench.py --steps 300 --cpu-rounds 0 --slots $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('slots=$s rep=$rep', round(d['value']), round(d['ms_per_step'],3), d['host_cpu'])"
done; done
