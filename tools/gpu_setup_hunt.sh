#!/bin/bash
# some processes show 0.3-0.5 s of job set-up instead of 0.06 s: print the set-up marks of such a run
for i in 1 2 3 4 5 6 7 8; do
  DPH_PROFILE=1 DP_ALLOC_TRACE=1 timeout 300 python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 > gpurun_out/hunt.json 2> gpurun_out/hunt.err
  s=$(python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/hunt.json') if l.startswith('{')][-1])
print('%.3f' % d['job_breakdown_s']['setup_value_table_kmer_index_slots'])")
  echo "run $i setup $s"
  if python3 -c "import sys; sys.exit(0 if float('$s') > 0.15 else 1)"; then
    grep -E "^\[setup\]" gpurun_out/hunt.err | tail -14
    grep -E "^\[alloc\]" gpurun_out/hunt.err | awk '{ if ($(NF-1)+0 > 2.0) print }' | tail -20
    break
  fi
done
