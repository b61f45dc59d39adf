#!/bin/bash
R=gpurun_out/r06; mkdir -p $R
COMMON="--k 10 --steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --max-rounds 60 --slots 1"
DP_LIB_DIR=$PWD/downpore_amd/lib_prof DP_CHAIN_PROF=1 timeout 600 python3 bench.py $COMMON 2>&1 >/dev/null | grep 'final walk' | sed -n '20,60p'
