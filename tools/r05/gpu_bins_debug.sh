#!/bin/bash
# diagnosis of the binned counting step: how full the bins are, round by round (one slot, first rounds of config 2; then k = 10)
R=gpurun_out/r05; mkdir -p $R
C="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --slots 1"
DP_KX_BINS_DEBUG=1 DP_KX_ONESHOT_DEBUG=1 timeout 300 python3 bench.py $C --max-rounds 12 > $R/binsdbg_k13.json 2> $R/binsdbg_k13.err; echo "k13 rc $?"
grep -E "kx bins|one-go|fault|error" $R/binsdbg_k13.err | head -30
DP_KX_BINS_DEBUG=1 DP_KX_ONESHOT_DEBUG=1 timeout 300 python3 bench.py $C --k 10 --max-rounds 6 > $R/binsdbg_k10.json 2> $R/binsdbg_k10.err; echo "k10 rc $?"
grep -E "kx bins|one-go|fault|error" $R/binsdbg_k10.err | head -20
tail -3 $R/binsdbg_k10.err | cut -c1-300
