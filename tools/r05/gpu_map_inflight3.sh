#!/bin/bash
# mapper threads x reads in flight per thread again, now that a context's teardown costs 0.4 ms (config 3, 6 runs each)
R=gpurun_out/r05; mkdir -p $R
for cfg in "4 2730" "6 1820" "6 2730" "8 1365" "8 2048" "5 2184" "4 2730" "6 1820"; do
set -- $cfg
export DP_MAP_THREADS=$1 DP_MAP_INFLIGHT=$2
timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 6 --map-cpu-baseline 0 > $R/map_marks.json 2> /dev/null
python3 -c "
import json
d=json.loads([l for l in open('$R/map_marks.json') if l.startswith('{')][-1])['map_config3']; print('threads $1 inflight $2: %.0f reads/s' % d['value'], [round(x,4) for x in d['wall_s_runs']], {k:round(v,4) for k,v in d['breakdown_s'].items()}, 'kernel ms', round(d['map_kernel']['ms_total'],1), 'launches', d['map_kernel']['launches'], d['paf_sha256_matches_oracle_fixture'])"
done
