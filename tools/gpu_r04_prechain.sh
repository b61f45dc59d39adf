#!/bin/bash
# round 4: the chunk stage launched behind the un-waited scan (dp_index_prechain) - parity tests, then A/B
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -x -q -m gpu > gpurun_out/r04/prechain_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" gpurun_out/r04/prechain_tests.log | tail -2
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r04/prechain_full.log 2>&1; echo "full-size rc $?"; tail -2 gpurun_out/r04/prechain_full.log
REPS=3 timeout 900 python3 tools/ab.py off:.:DP_INDEX_PRECHAIN=0 on:.: 2>&1 | tee gpurun_out/r04/ab_prechain.txt
