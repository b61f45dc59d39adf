#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -x -q -m gpu > gpurun_out/r04/rows_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" gpurun_out/r04/rows_tests.log | tail -2
REPS=3 NAME=rows VARIANTS="atomic:.:DP_INDEX_FILL_ATOMIC=1 rows:.:" tools/gpu_r04_ab.sh
TAG=_rows tools/gpu_r04_timeline.sh | grep -E "chunk_kernel|index_fill|posting|kernels "
