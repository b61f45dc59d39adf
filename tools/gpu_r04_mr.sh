#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_overlap_e2e.py tests/test_gpu_bench.py -x -q -m gpu -k "round or rank or shard or plain_launch or layouts" > gpurun_out/r04/mr_tests.log 2>&1; echo "rc $?"; grep -E "passed|failed" gpurun_out/r04/mr_tests.log | tail -2
