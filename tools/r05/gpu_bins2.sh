#!/bin/bash
# round 5, second attempt: parity of the binned counting step (incl. overflowing bins), full-size jobs, then the A/B and the timelines
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_overlap_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "counting_step or kmer_index or index_mode or sort_tiers or scan_prepare or compaction" > $R/bins_tests.log 2>&1; echo "tests rc $?"; tail -3 $R/bins_tests.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "config2 or flagged or config4" > $R/bins_full.log 2>&1; echo "full-size rc $?"; tail -3 $R/bins_full.log
REPS=${REPS:-3} timeout 1200 python3 tools/ab.py records:.:DP_KX_BINS=0 bins:.:DP_KX_BINS=1,DP_KX_ONESHOT_DEBUG=1 2>&1 | tee $R/ab_bins.txt
C="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --slots 1"
for v in 0 1; do
  rm -rf $R/kt1
  DP_KX_BINS=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py $C --max-rounds 120 > $R/kt1.json 2> $R/kt1.err; echo "trace rc=$?"
  t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
  python3 tools/round_timeline.py $t > $R/round_timeline_one_slot_bins$v.txt; head -8 $R/round_timeline_one_slot_bins$v.txt; tail -1 $R/round_timeline_one_slot_bins$v.txt
  rm -rf $R/kt1
done
for v in 0 1; do
  DP_KX_BINS=$v timeout 600 python3 bench.py --steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 12 --dense-job 0 --map-leg-repeats 0 2> $R/dense_bins$v.err > $R/dense_bins$v.json
  python3 - <<PY
import json
d=json.loads([l for l in open('$R/dense_bins$v.json') if l.startswith('{')][-1])
print('bins=$v value %.2fM rounds_only %.4f' % (d['value']/1e6, d['rounds_only']['ms_per_round']), {k:round(v,4) for k,v in d['kernel_ms_per_round'].items()})
for key in ('index_query_dense', 'index_query_dense_slots'):
    l=d.get(key) or {}
    print('  ', key, 'ms/round', l.get('ms_per_round'), l.get('kernel_ms_per_round'), (l.get('parity') or {}).get('paf_sha256_matches_oracle_fixture'), (l.get('query_kernel') or {}).get('frac_of_hbm_peak'))
PY
done
rm -f $R/kt1.json $R/kt1.err $R/dense_bins*.err
