#!/bin/bash
# sub-phase stamps of the consensus kernel's contig + PAF phase (lib_p5: -DCF_P5PROF), then a driver-style bench with the longest step waits per job
R=gpurun_out/r05; mkdir -p $R
DP_LIB_DIR=$PWD/downpore_amd/lib_p5 DP_CONS_DEBUG=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 40 --slots 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/cons_p5.txt > /dev/null
grep "contig+paf, us" $R/cons_p5.txt | tail -4
grep "slowest window" $R/cons_p5.txt | tail -3
timeout 600 python3 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2>/dev/null > $R/bench_per_job_parts.json
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/bench_per_job_parts.json') if l.startswith('{')][-1])
print('value %.2fM ms/step %.1f rounds_only %.4f parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['parity']['paf_sha256_matches_oracle_fixture']))
jb=d['job_breakdown_s']
for t,p in zip(jb['per_job'], jb['per_job_ms_setup_waitplan_waitfmt_commitidle_longest_step_waits']): print('  job %.1f ms | setup %.1f wait-plan %.1f wait-fmt %.1f commit-idle %.1f | longest waits (ms, at round) %s' % (t*1e3, *p[:4], p[4]))
PY
