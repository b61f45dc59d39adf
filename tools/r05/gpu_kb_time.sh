#!/bin/bash
# set-up time + index-build kernel times of the current build (no tests)
R=gpurun_out/r05; mkdir -p $R
for v in after after; do
  timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v: value %.2fM ms/step %.1f rounds_only %.4f setup %.2f ms parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], 1e3*d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity']['paf_sha256_matches_oracle_fixture']))"
done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf $R/kb_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kb_trace -- python3 bench.py --steps 3 --warmup 1 --max-rounds 20 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 > /dev/null 2>&1
f=$(find $R/kb_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "kb_|values|Name" $f | cut -c1-200 > $R/kb_kernel_stats_after.csv; cat $R/kb_kernel_stats_after.csv | sed 's/(.*)//' | cut -c1-120 | head -14
rm -rf $R/kb_trace
