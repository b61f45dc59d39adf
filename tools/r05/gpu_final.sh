#!/bin/bash
# the driver's checks at the end of the round: the whole -m gpu suite, smoke() three times, the driver-style bench line
R=gpurun_out/r05; mkdir -p $R
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=8 > $R/gpu_suite_final.log 2>&1; echo "suite rc $?"; grep -E "passed|failed|error" $R/gpu_suite_final.log | tail -3
for i in 1 2 3; do timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/smoke_$i.log 2>&1; echo "smoke $i rc $? $(tail -1 $R/smoke_$i.log)"; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_driver_style.json 2> $R/bench_driver_style.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/bench_driver_style.json') if l.startswith('{')][-1])
print('value %.2fM ms/step %.1f rounds_only %.4f setup %.4f parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity']))
print('roofline', d['roofline'])
print('cpu_baseline', d.get('cpu_baseline'))
print('ground_truth', d.get('ground_truth'))
for k in ('index_query_dense','index_query_dense_slots','overlap_default_k10_job','map_config3'):
    print(k, json.dumps(d.get(k))[:1500])
PY
