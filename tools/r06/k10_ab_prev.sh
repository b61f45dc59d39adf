#!/bin/bash
# whole k = 10 config-2 jobs, alternating: the build of the end of round 5 (_ab/prev) against the working tree
R=gpurun_out/r06; mkdir -p $R
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for rep in 1 2 3; do
  for d in _ab/prev .; do
    for s in ${SLOTS:-5 6}; do
      (cd $d && timeout 300 python3 bench.py --k 10 --steps 2 --warmup 1 --slots $s $OFF 2>/dev/null) | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-8s slots $s job %.4f s ms/round %.4f'%('$d',j['job_breakdown_s']['whole_job'],j['rounds_only']['ms_per_round']),{k:round(v,3) for k,v in j['kernel_ms_per_round'].items()})" | tee -a $R/k10_ab_prev_${TAG:-x}.txt
    done
  done
done
