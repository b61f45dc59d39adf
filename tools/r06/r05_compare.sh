#!/bin/bash
# k = 13 config-2 jobs: the build of the end of round 5 (_ab/prev) against the working tree - kernel and phase times per round, alternating
for rep in 1 2 3; do for d in _ab/prev .; do
(cd $d && timeout 300 python3 bench.py --steps 4 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 2>/dev/null) | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-9s %.2f M rounds %.4f ms | kernels'%('$d', j['value']/1e6, j['rounds_only']['ms_per_round']), {k:round(v,4) for k,v in j['kernel_ms_per_round'].items()}, '| phases', {k:round(v,4) for k,v in j['phase_ms_per_round'].items()})"
done; done
