#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -x -q -m gpu > gpurun_out/r04/fast_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed" gpurun_out/r04/fast_tests.log | tail -1
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "config2 or config4 or flag or variable" 2>&1 | grep -E "passed|failed"
REPS=4 NAME=fast VARIANTS="items:.:DP_KX_FAST=0 fast:.:" tools/gpu_r04_ab.sh
TAG=_fast tools/gpu_r04_timeline.sh | grep -E "kidx_walk|kernels " | tail -2
for v in 0 1; do DP_KX_FAST=$v ROUNDS=300 tools/gpu_r04_config4.sh | head -3 | tail -2; done
