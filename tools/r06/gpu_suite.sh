#!/bin/bash
# the driver's GPU checks: the whole -m gpu suite (with durations), then smoke() twice, then the driver-style bench
R=gpurun_out/r06; mkdir -p $R
timeout 2700 python3 -m pytest tests/ -x -q -m gpu --durations=15 > $R/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed|error" $R/gpu_suite.log | tail -3; grep -A18 'slowest' $R/gpu_suite.log | head -20
for i in 1 2; do python3 -c "import __graft_entry__ as g; g.smoke()" > $R/smoke_$i.log 2>&1; echo "smoke $i rc $? $(tail -1 $R/smoke_$i.log)"; done
if [ -n "$BENCH" ]; then timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/bench_driver_style_${TAG:-x}.json 2> $R/bench_driver_style.err; echo "bench rc $?"; python3 -c "
import json
j=json.loads([l for l in open('$R/bench_driver_style_${TAG:-x}.json') if l.startswith('{')][-1])
print('value %.2f M  ms/step %.2f  rounds_only %.4f ms  parity %s  roofline %s'%(j['value']/1e6,j['ms_per_step'],j['rounds_only']['ms_per_round'],j['parity'],{k:j['roofline'][k] for k in ('frac','achieved','launch_ms')}))
d=j['overlap_default_k10_job']; print('k10 job %.3f s'%d['wall_s'], d['parity'], {k:round(v['frac_of_hbm_peak'],4) for k,v in d['kernels_per_round'].items()})
m=j['map_config3']; print('map', m['value'], m['wall_s_runs'], m['paf_sha256_matches_oracle_fixture'])"; fi
