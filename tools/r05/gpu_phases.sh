#!/bin/bash
# round 5: what do the kernels of five concurrent rounds wait for?  The per-wave phase timers of the PROF build (every phase ends in
# s_waitcnt vmcnt(0) lgkmcnt(0) and a clock read) for the chaining kernels and the count walk, with ONE round in flight and with FIVE:
# phases made of loads (records, stage a, prefilter, stage b) against phases made of LDS + ALU work (initial, events + walk).
R=gpurun_out/r05; mkdir -p $R
COMMON="--steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --max-rounds 240"
for s in 1 5; do
  DP_KX_BINS=0 DP_LIB_DIR=$PWD/downpore_amd/lib_prof DP_CHAIN_PROF=1 timeout 600 python3 bench.py $COMMON --slots $s > $R/phases_s$s.json 2> $R/phases_s$s.err; echo "slots $s rc $?"
  python3 tools/r05/chainprof_digest.py $R/phases_s$s.err 40 | tee $R/chain_phases_slots$s.txt
  python3 - <<PY
import json
d=json.loads([l for l in open('$R/phases_s$s.json') if l.startswith('{')][-1])
print('slots $s: ms/round %.4f' % d['rounds_only']['ms_per_round'], {k:round(v,4) for k,v in d['kernel_ms_per_round'].items()})
PY
  rm -f $R/phases_s$s.err
done
