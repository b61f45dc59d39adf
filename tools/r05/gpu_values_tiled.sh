#!/bin/bash
# value pass with the reverse complement's count out of LDS tiles instead of a gather: parity, then set-up and kernel times
R=gpurun_out/r05; mkdir -p $R
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py tests/test_golden.py -x -q -m gpu -k "value or paf_bit_exact or golden or full_run" > $R/values_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/values_tests.log
bash tools/r05/gpu_kb_time.sh
