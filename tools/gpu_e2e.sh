#!/bin/bash
# run each e2e test separately under a hard timeout; logs to gpurun_out/
make -C oracle -j8 >/dev/null 2>&1
mkdir -p gpurun_out
for t in "test_overlap_paf_bit_exact[10-100000-400-5000-0.0-False]" "test_overlap_paf_bit_exact[10-80000-300-6000-0.03-True]" "test_overlap_paf_bit_exact[13-1500000-3000-10000-0.0-False]" "test_overlap_paf_bit_exact[13-1200000-2000-12000-0.002-True]" test_overlap_full_run_config1_k10 test_overlap_himem_false_top_level_reads test_values_table_matches_oracle test_cli_matches_oracle_cli; do
  echo "=== $t" >> gpurun_out/e2e.log
  timeout -s KILL 150 python -m pytest "tests/test_gpu_overlap_e2e.py::$t" -m gpu -x -q --timeout=120 --timeout-method=thread >> gpurun_out/e2e.log 2>&1
  echo "rc=$?" >> gpurun_out/e2e.log
done
tail -c 6000 gpurun_out/e2e.log
