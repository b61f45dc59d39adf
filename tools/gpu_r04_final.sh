#!/bin/bash
# end-of-round checks as the driver runs them: the whole -m gpu suite, smoke() three times, the driver-style bench line
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed|error" gpurun_out/r04/gpu_suite.log | tail -3
for i in 1 2 3; do python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke_$i.log 2>&1; echo "smoke $i rc $? $(tail -1 gpurun_out/r04/smoke_$i.log)"; done
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_style_steps20_warmup5.json 2> gpurun_out/r04/bench_driver_style.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/bench_driver_style_steps20_warmup5.json') if l.startswith('{')][-1])
print('value %.2fM ms/step %.1f rounds_only %.4f setup %.4f parity %s incl_upload %.2fM' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'], d['parity'], d['value_incl_upload']/1e6))
print('roofline', d['roofline']['frac'], d['roofline']['launch_ms'], 'dense', d['index_query_dense']['ms_per_round'], d['index_query_dense']['query_kernel']['frac_of_hbm_peak'], d['index_query_dense']['parity'])
print('map', d['map_config3']['value'], d['map_config3']['paf_sha256_matches_oracle_fixture'], 'gt', d.get('ground_truth'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
PY
