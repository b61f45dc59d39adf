#!/bin/bash
# `map`, config 3, 12 runs per setting in one process: the block cache at its new cap (16 GB), reads travelling while they are mapped or not
R=gpurun_out/r05; mkdir -p $R
for a in 0 1 0 1; do
export DP_MAP_ASYNC_UPLOAD=$a
timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 12 --map-cpu-baseline 0 > $R/map_marks.json 2> /dev/null
python3 -c "
import json
d=json.loads([l for l in open('$R/map_marks.json') if l.startswith('{')][-1])['map_config3']; w=d['wall_s_runs']; print('async upload $a: best %.0f reads/s, mean of runs 2.. %.1f ms' % (d['value'], 1e3*sum(w[1:])/len(w[1:])), [round(x,4) for x in w], {k:round(v,4) for k,v in d['breakdown_s'].items()}, d['paf_sha256_matches_oracle_fixture'])"
done
