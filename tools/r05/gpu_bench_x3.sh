#!/bin/bash
# the driver's bench command three times on one box (last code state)
R=gpurun_out/r05; mkdir -p $R
for i in 1 2 3; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_final_run$i.json 2>/dev/null
python3 - $i <<'PY'
import json,sys
i=sys.argv[1]
d=json.loads([l for l in open('gpurun_out/r05/bench_final_run%s.json' % i) if l.startswith('{')][-1])
pj=sorted(d['job_breakdown_s']['per_job'])
print('run %s: value %.2fM ms/step %.1f (median job %.1f, fastest %.1f) rounds_only %.4f setup %.2f ms parity %s | map %.0f k reads/s | k10 job %.3f s | dense query %.1f us' % (i, d['value']/1e6, d['ms_per_step'], 1e3*pj[len(pj)//2], 1e3*pj[0], d['rounds_only']['ms_per_round'], 1e3*d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity']['paf_sha256_matches_oracle_fixture'], d['map_config3']['value']/1e3, d['overlap_default_k10_job']['wall_s'], 1e3*d['index_query_dense']['query_kernel']['launch_ms']))
PY
done
