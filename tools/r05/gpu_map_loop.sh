#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 3 --map-cpu-baseline 0 > $R/map_marks.json 2> $R/map_marks.err
grep -E "map loop" $R/map_marks.err | tail -6
rm -f $R/map_marks.err $R/map_marks.json
