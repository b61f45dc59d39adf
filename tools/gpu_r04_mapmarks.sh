#!/bin/bash
mkdir -p gpurun_out/r04
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 4 > gpurun_out/r04/map_marks.json 2> gpurun_out/r04/map_marks.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04/map_marks.json') if l.startswith('{')][-1])['map_config3']; print(d['value'], d['wall_s_runs'], d['breakdown_s'], d['paf_sha256_matches_oracle_fixture'])"
grep "map setup" gpurun_out/r04/map_marks.err | tail -7
