#!/bin/bash
# round 5: the counting step with hits binned by read range (kidx_walk_bin / kidx_bin_count / kidx_bin_fill) - parity, then A/B against
# round 4's hit records (DP_KX_BINS=0), the one-slot timeline of a round in both forms, and the dense leg
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -x -q -m gpu > $R/bins_tests.log 2>&1; echo "tests rc $?"; tail -3 $R/bins_tests.log
python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "config2 or flagged or config4_first" > $R/bins_full.log 2>&1; echo "full-size rc $?"; tail -3 $R/bins_full.log
REPS=${REPS:-3} python3 tools/ab.py records:.:DP_KX_BINS=0 bins:.:DP_KX_BINS=1,DP_KX_ONESHOT_DEBUG=1 2>&1 | tee $R/ab_bins.txt
for v in 0 1; do
  rm -rf $R/kt1
  DP_KX_BINS=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py --steps 1 --warmup 0 --max-rounds 120 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots 1 > $R/kt1.json 2> $R/kt1.err; echo "trace rc=$?"
  t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
  python3 tools/round_timeline.py $t > $R/round_timeline_one_slot_bins$v.txt; tail -30 $R/round_timeline_one_slot_bins$v.txt
  rm -rf $R/kt1
done
for v in 0 1; do
  DP_KX_BINS=$v DP_KX_ONESHOT_DEBUG=1 python3 bench.py --steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 12 --map-leg-repeats 0 2> $R/dense_bins$v.err > $R/dense_bins$v.json
  python3 - <<PY
import json
d=json.loads([l for l in open('$R/dense_bins$v.json') if l.startswith('{')][-1])
print('bins=$v value %.2fM rounds_only %.4f' % (d['value']/1e6, d['rounds_only']['ms_per_round']), {k:round(v,4) for k,v in d['kernel_ms_per_round'].items()})
print('  dense leg:', json.dumps(d.get('index_query_dense'))[:900])
PY
  grep -c "one-go step repeated" $R/dense_bins$v.err
done
