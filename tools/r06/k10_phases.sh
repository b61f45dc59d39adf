#!/bin/bash
# Round 6: per-wave phase timers of the chaining kernels in the dense regime (k = 10), one slot: where do chain_walk<0>'s 400 us go?
R=gpurun_out/r06; mkdir -p $R
COMMON="--k 10 --steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --max-rounds 60"
for s in 1; do
  DP_LIB_DIR=$PWD/downpore_amd/lib_prof DP_CHAIN_PROF=1 timeout 600 python3 bench.py $COMMON --slots $s > $R/k10_phases_s$s.json 2> $R/k10_phases_s$s.err; echo "slots $s rc $?"
  python3 tools/r05/chainprof_digest.py $R/k10_phases_s$s.err 10 | tee $R/k10_chain_phases_slots$s.txt
  grep 'chain prof' $R/k10_phases_s$s.err | sed -n '200,224p'
  grep -c 'chain prof' $R/k10_phases_s$s.err
  rm -f $R/k10_phases_s$s.err
done
