#!/bin/bash
for rep in 1 2; do for w in 8 100; do
timeout 300 python bench.py --steps 200 --warmup $w --cpu-rounds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); a=d.get('scan_kernels_leg') or d.get('index_mode'); print('warmup=$w rep=$rep main', d['scan_mode'][:8], round(d['value']), round(d['ms_per_step'],3), 'alt', round(a['value']), round(a['ms_per_step'],3), d['host_cpu'])"
done; done
