#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
C="--gpus 1 --steps 12 --warmup 3 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for s in 5 6 7 4 5 6; do
  timeout 600 python bench.py $C --slots $s > $R/slots.json 2>/dev/null
  python3 - $s <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/slots.json') if l.startswith('{')][-1])
pj=d['per_rank'][0]['per_job']
print('slots %s: value %.2fM rounds_only %.4f | slots waited for plans %.1f ms, lanes %s' % (sys.argv[1], d['value']/1e6, d['rounds_only']['ms_per_round'], pj['slot_wait_for_plan_us']/1e3, d['per_rank'][0].get('planner_lanes_at_the_end')))
PY
done
rm -f $R/slots.json
for i in 1 2; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_driver_style_$i.json 2> /dev/null; echo "bench rc $?"
python3 - $i <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/r05/bench_driver_style_%s.json' % sys.argv[1]) if l.startswith('{')][-1])
print('driver-style %s: value %.2fM ms/step %.1f rounds_only %.4f setup %.4f parity %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity']['paf_sha256_matches_oracle_fixture']))
print('   k10 job %.2fM overlaps/s (%.3f s), map %.0f reads/s, dense query frac %.3f / %.3f (5 slots)' % (d['overlap_default_k10_job']['value']/1e6, d['overlap_default_k10_job']['wall_s'], d['map_config3']['value'], d['index_query_dense']['query_kernel']['frac_of_hbm_peak'], d['index_query_dense_slots']['query_kernel']['frac_of_hbm_peak']))
PY
done
