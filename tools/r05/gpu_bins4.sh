#!/bin/bash
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "counting_step or index_mode" > $R/bins_tests3.log 2>&1; echo "tests rc $?"; tail -2 $R/bins_tests3.log
C="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --slots 1"
for v in "DP_KX_BIN_WAVES=8" "DP_KX_BIN_WAVES=4"; do
  rm -rf $R/kt1; export $v
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py $C --max-rounds 120 > $R/kt1.json 2> $R/kt1.err; echo "trace $v rc=$?"
  unset DP_KX_BINS DP_KX_BIN_WAVES
  t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
  for r in 30 50 70 90; do python3 tools/round_timeline.py $t $r | grep -E "kidx_walk|kernels "; done
  python3 tools/round_timeline.py $t 60 > "$R/round_timeline_one_slot_tail_$v.txt"
  rm -rf $R/kt1
done
REPS=${REPS:-4} timeout 1500 python3 tools/ab.py records:.:DP_KX_BINS=0 bins8:.:DP_KX_BIN_WAVES=8 bins4:.:DP_KX_BIN_WAVES=4 2>&1 | grep -v "committing\|host:" | tee $R/ab_bins_tail.txt
rm -f $R/kt1.json $R/kt1.err
