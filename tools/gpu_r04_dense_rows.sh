#!/bin/bash
mkdir -p gpurun_out/r04
for v in 0 1; do
DP_INDEX_FILL_ROWS=$v timeout 600 python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds 12 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots 1 > gpurun_out/r04/dense_rows$v.json 2> gpurun_out/r04/dense_rows$v.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/dense_rows$v.json') if l.startswith('{')][-1]); print('rows=$v ms/round %.3f' % d['rounds_only']['ms_per_round'], d['phase_ms_per_round'])"
done
timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "k10" 2>&1 | grep -E "passed|failed"
DP_INDEX_FILL_ROWS=1 timeout 600 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "k10" 2>&1 | grep -E "passed|failed"
