#!/bin/bash
# A/B of one environment switch on whole config-2 jobs: tools/gpu_ab_env.sh VAR "v1 v2 ..." [bench args]
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/r03
for rep in 1 2; do
for v in $VALS; do
  env $VAR=$v python3 bench.py --steps ${STEPS:-8} --warmup 2 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); b=d['job_breakdown_s']
print('$VAR=$v', 'value %.2f M' % (d['value']/1e6), 'ms/job %.1f' % d['ms_per_step'], 'rounds %.1f ms' % (1e3*b['rounds']), 'setup %.1f ms' % (1e3*b['setup_value_table_kmer_index_slots']))"
done; done
