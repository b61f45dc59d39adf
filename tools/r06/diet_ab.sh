#!/bin/bash
# launch diet (scan in walk 0, third pass sized by the previous round, later passes on the full layout): parity subset, then alternating
# runs against the previous commit's build (_ab/prev) at k = 13, and the k = 10 job with the later passes slim / full
R=gpurun_out/r06; mkdir -p $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -m gpu -x -q -k "scan_index_query_chain or chain_shortcuts or paf_bit_exact or config1_k10 or other_query_types or identical_and_repetitive or degenerate" 2>&1 | tail -3
REPS=4 timeout 1500 python3 tools/ab.py prev:_ab/prev: now:.: full9:.:DP_CHAIN_FULL_FROM=9 2>&1 | tee $R/ab_launch_diet_k13.txt
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for v in 1 9 1 9; do
  for s in 1 5; do
    DP_CHAIN_FULL_FROM=$v timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('k10 full_from $v slots $s job %.4f s ms/round %.4f'%(j['job_breakdown_s']['whole_job'],j['rounds_only']['ms_per_round']),{k:round(v,3) for k,v in j['kernel_ms_per_round'].items()})" | tee -a $R/ab_full_from_k10.txt
  done
done
