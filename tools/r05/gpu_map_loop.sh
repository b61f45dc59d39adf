#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
for cfg in "3 0" "4 2730"; do
set -- $cfg
export DP_MAP_THREADS=$1
if [ $2 = 0 ]; then unset DP_MAP_INFLIGHT; else export DP_MAP_INFLIGHT=$2; fi
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 3 --map-cpu-baseline 0 > $R/map_marks.json 2> $R/map_marks_$1.err
echo "== threads $1 inflight $2"
grep -E "^\[map (thread|end|loop)" $R/map_marks_$1.err | tail -12
done
