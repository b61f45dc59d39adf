#!/bin/bash
# the scan-kernel leg (DP_SCAN_INDEX=0 rounds) of bench.py: round-5 build against the working tree, alternating
for rep in 1 2 3; do for d in _ab/prev .; do
(cd $d && timeout 300 python3 bench.py --steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 60 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 2>/dev/null) | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['scan_kernels_leg']
print('%-9s scan leg %.4f ms per round'%('$d', s['ms_per_round']), {k:round(v,4) for k,v in s['kernel_ms_per_round'].items()})"
done; done
