#!/bin/bash
# whole-job bench with the host-side profile (set-up marks, per-section CPU), then the GPU test suite
mkdir -p gpurun_out
DPH_PROFILE=1 timeout 900 python3 bench.py --steps ${STEPS:-3} --warmup 1 --cpu-rounds 0 ${BENCH_ARGS} > gpurun_out/job_bench.json 2> gpurun_out/job_bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/job_bench.json
grep -E "^\[setup\]|^\[prof\]|rounds|plan" gpurun_out/job_bench.err | tail -60
if [ -n "$RUN_TESTS" ]; then timeout 1500 python3 -m pytest tests -m gpu -x -q ${TEST_ARGS} 2>&1 | tail -15; fi
