#!/bin/bash
# k = 10: existing knobs of the index step (count as a launch of its own, waves per walk workgroup) + the scan kernels instead of the index
R=gpurun_out/r06; mkdir -p $R
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
run() { # label env...
  local label=$1; shift
  for s in 1 5; do
    env "$@" timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-28s slots $s job %.4f s ms/round %.4f'%('$label',j['job_breakdown_s']['whole_job'],j['rounds_only']['ms_per_round']),{k:round(v,3) for k,v in j['kernel_ms_per_round'].items()})" | tee -a $R/k10_knobs.txt
  done
}
run default A=1
run count_own_launch DP_KX_FUSE=0
run walk_waves16 DP_KX_BIN_WAVES=16
run walk_waves4 DP_KX_BIN_WAVES=4
run count_own+waves16 DP_KX_FUSE=0 DP_KX_BIN_WAVES=16
run scan_kernels DP_SCAN_INDEX=0
run default_again A=1
