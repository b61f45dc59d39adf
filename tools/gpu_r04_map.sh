#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_map.py -x -q -m gpu > gpurun_out/r04/map_tests.log 2>&1; echo "map tests rc $?"; grep -E "passed|failed" gpurun_out/r04/map_tests.log | tail -2
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "map or config3 or config5" > gpurun_out/r04/map_full.log 2>&1; echo "map full rc $?"; grep -E "passed|failed" gpurun_out/r04/map_full.log | tail -2
for v in 1 0; do DP_MAP_ONE_LANE=$v timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 4 > gpurun_out/r04/map_bench_$v.json 2> gpurun_out/r04/map_bench_$v.err; python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/map_bench_$v.json') if l.startswith('{')][-1])['map_config3']; print('one_lane=$v', d['value'], d['wall_s_runs'], d['map_kernel'], d['breakdown_s'], d['paf_sha256_matches_oracle_fixture'])"; done
