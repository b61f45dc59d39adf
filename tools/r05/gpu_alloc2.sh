#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 4 --map-cpu-baseline 0 > $R/map_marks.json 2> $R/map_marks.err
grep -E "^\[map (thread|end)" $R/map_marks.err | tail -12
for i in 1 2; do
timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 6 --map-cpu-baseline 0 > $R/map_marks.json 2> /dev/null
python3 -c "
import json
d=json.loads([l for l in open('$R/map_marks.json') if l.startswith('{')][-1])['map_config3']; print('%.0f reads/s' % d['value'], [round(x,4) for x in d['wall_s_runs']], {k:round(v,4) for k,v in d['breakdown_s'].items()}, d['paf_sha256_matches_oracle_fixture'])"
done
