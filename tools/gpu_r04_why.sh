#!/bin/bash
mkdir -p gpurun_out/r04
DP_CONS_WHY=1 timeout 600 python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds 8 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots 1 > gpurun_out/r04/dense_why.json 2> gpurun_out/r04/dense_why.err; echo "rc $?"
grep "\[cons\]" gpurun_out/r04/dense_why.err | head -20
