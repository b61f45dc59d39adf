#!/bin/bash
# consensus kernel, contig + PAF phase rewritten (prefix sums of the consensus' gaps, bisection, four pairs a trip): parity, then the
# phase timers of a 40-round one-slot job before (lib_before = the commit before) and after, then a driver-style bench with per-job parts
R=gpurun_out/r05; mkdir -p $R
timeout 1500 python -m pytest tests/test_golden.py tests/test_gpu_overlap_e2e.py tests/test_gpu_full_size.py tests/test_gpu_flag_matrix.py -x -q -m gpu -k "golden or consensus or paf_bit_exact or full_run or config2 or (flags and not map)" > $R/cons_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/cons_tests.log
digest() {
python3 - "$1" <<'PY'
import re, sys, statistics as st
L = open(sys.argv[1]).read().splitlines()
def med(pat, n):
    rows = [list(map(float, re.findall(pat, l)[0])) for l in L if re.findall(pat, l)]
    rows = rows[5:]
    return [st.median(r[i] for r in rows) for i in range(n)] if rows else None
m = med(r"gather\+query ([\d.]+)/([\d.]+) trim ([\d.]+)/([\d.]+) shared\+reduce ([\d.]+)/([\d.]+) align ([\d.]+)/([\d.]+) contig\+paf ([\d.]+)/([\d.]+) \| group total ([\d.]+)/([\d.]+)", 12)
print("mean/max per phase (median over rounds): gather %s/%s trim %s/%s shared+reduce %s/%s align %s/%s contig+paf %s/%s total %s/%s" % tuple(m) if m else "no phase lines")
PY
}
for v in before after before after; do
  if [ $v = before ]; then export DP_LIB_DIR=$PWD/downpore_amd/lib_before; else unset DP_LIB_DIR; fi
  DP_CONS_DEBUG=1 timeout 600 python3 bench.py --steps 1 --warmup 0 --max-rounds 40 --slots 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2> $R/cons_debug_$v.txt > /dev/null
  echo "$v: $(digest $R/cons_debug_$v.txt)"
  timeout 600 python3 bench.py --steps 3 --warmup 1 --slots 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   one slot: ms/round %.4f parity %s k_cons_ms per job %.2f' % (j['rounds_only']['ms_per_round'], j['parity']['paf_sha256_matches_oracle_fixture'], j['per_rank'][0]['kernel_ms_per_job']['k_cons_ms']))"
done
unset DP_LIB_DIR
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --dense-job 0 2>/dev/null > $R/bench_per_job_parts.json
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05/bench_per_job_parts.json') if l.startswith('{')][-1])
print('value %.2fM ms/step %.1f rounds_only %.4f parity %s' % (d['value']/1e6, d['ms_per_step'], d['rounds_only']['ms_per_round'], d['parity']['paf_sha256_matches_oracle_fixture']))
jb=d['job_breakdown_s']
for t,p in zip(jb['per_job'], jb['per_job_ms_setup_waitplan_waitfmt_commitidle']): print('  job %.1f ms | setup %.1f wait-plan %.1f wait-fmt %.1f commit-idle %.1f' % (t*1e3, *p))
PY
