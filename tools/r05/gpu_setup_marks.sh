#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
DPH_PROFILE=1 timeout 600 python3 bench.py --gpus 1 --steps 3 --warmup 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/setup_marks.txt > /dev/null
grep "^\[setup\]" $R/setup_marks.txt | tail -12
DPH_START_TRACE=1 timeout 600 python3 bench.py --gpus 1 --steps 2 --warmup 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/start_trace.txt > /dev/null
grep "^\[start\]" $R/start_trace.txt | tail -12
