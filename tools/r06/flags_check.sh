#!/bin/bash
# the reads' flags on the device from a pinned shadow: parity of the overlap suites that flag reads, then start-up timelines and a short bench
R=gpurun_out/r06; mkdir -p $R
timeout 1800 python3 -m pytest tests/test_gpu_overlap_e2e.py tests/test_golden.py tests/test_planner_epoch.py -x -q -m gpu 2>&1 | tail -3
bash tools/r06/step_timeline.sh 2>&1 | grep "^job" | tee $R/job_start_after.txt
bash tools/r06/bench_quick.sh 2>&1 | grep -E "k13 value|k10 job"
