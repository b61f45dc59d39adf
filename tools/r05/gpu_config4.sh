#!/bin/bash
# config-4 scale on ONE GPU (1 M reads x 10 kb resident, 50 GB k-mer index): set-up and the first 600 rounds of a job, records against bins
R=gpurun_out/r05; mkdir -p $R
for v in "DP_KX_BINS=0" "DP_KX_BINS=1"; do
  export $v
  timeout 1200 python3 bench.py --reads 1000000 --steps 1 --warmup 0 --max-rounds 600 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 > $R/config4_$v.json 2> $R/config4.err; echo "$v rc $?"
  unset DP_KX_BINS
  python3 - "$R/config4_$v.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('  value %.2fM ms/round %.4f setup %.3f s rounds %d' % (d['value']/1e6, d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['config']['rounds_per_step']))
print('  ', {k:round(v,4) for k,v in d['kernel_ms_per_round'].items()})
PY
done
rm -f $R/config4.err
