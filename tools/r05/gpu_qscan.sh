#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
timeout 1200 python -m pytest tests/test_gpu_overlap_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "paf_bit_exact or full_run or scan_index_query or ladder or index_mode" > $R/qscan_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/qscan_tests.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "config2 or flagged or config4_first" > $R/qscan_full.log 2>&1; echo "full-size rc $?"; tail -2 $R/qscan_full.log
REPS=3 tools/r05/ab12.sh DP_QUERY_SCAN=0 DP_QUERY_SCAN=1 | tee $R/ab12_query_scan.txt
