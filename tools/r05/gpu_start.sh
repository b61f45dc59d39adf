#!/bin/bash
# a job's first rounds against the end of its set-up (DPH_START_TRACE), three jobs; DP_PROF's set-up marks for one
R=gpurun_out/r05; mkdir -p $R
DPH_START_TRACE=1 timeout 600 python3 bench.py --gpus 1 --steps 3 --warmup 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/start_trace.txt > /dev/null
grep "^\[start\]" $R/start_trace.txt | tail -36
DP_PROF=1 timeout 600 python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/setup_marks.txt > /dev/null
grep "^\[setup\]" $R/setup_marks.txt | tail -12
