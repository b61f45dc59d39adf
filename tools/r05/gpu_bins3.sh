#!/bin/bash
# round 5: the binned counting step, workgroup sizes against round 4's records (five slots, whole jobs), clean one-slot timelines
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "counting_step" > $R/bins_tests2.log 2>&1; echo "tests rc $?"; tail -2 $R/bins_tests2.log
REPS=${REPS:-3} timeout 1500 python3 tools/ab.py records:.:DP_KX_BINS=0 bins4:.:DP_KX_BIN_WAVES=4 bins8:.:DP_KX_BIN_WAVES=8 bins16:.:DP_KX_BIN_WAVES=16 2>&1 | grep -v "committing\|host:" | tee $R/ab_bins_waves.txt
C="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --slots 1"
for v in "DP_KX_BINS=0" "DP_KX_BIN_WAVES=4" "DP_KX_BIN_WAVES=8" "DP_KX_BIN_WAVES=16"; do
  rm -rf $R/kt1
  env $v true
  export $v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py $C --max-rounds 120 > $R/kt1.json 2> $R/kt1.err; echo "trace $v rc=$?"
  unset DP_KX_BINS DP_KX_BIN_WAVES
  t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
  for r in 30 50 70 90; do python3 tools/round_timeline.py $t $r | grep -E "kidx_|kernels "; done > $R/timeline_kidx_$v.txt
  python3 tools/round_timeline.py $t 60 > $R/round_timeline_one_slot_$v.txt
  cat $R/timeline_kidx_$v.txt
  rm -rf $R/kt1
done
# the dense regime as whole jobs (k = 10 as the bench's main workload, one job each)
for v in "DP_KX_BINS=0" "DP_KX_BIN_WAVES=8" "DP_KX_BIN_WAVES=16"; do
  export $v
  timeout 600 python3 bench.py --k 10 --steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 2> /dev/null > $R/densejob_$v.json
  unset DP_KX_BINS DP_KX_BIN_WAVES
  python3 - <<PY
import json
d=json.loads([l for l in open('$R/densejob_$v.json') if l.startswith('{')][-1])
print('$v k=10 whole job: value %.2fM job %.3f s rounds_only %.4f ms' % (d['value']/1e6, d['job_breakdown_s']['whole_job'], d['rounds_only']['ms_per_round']), {k:round(x,4) for k,x in d['kernel_ms_per_round'].items()})
PY
done
rm -f $R/kt1.json $R/kt1.err
