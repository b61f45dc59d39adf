#!/bin/bash
# the driver's GPU checks: the whole -m gpu suite, then smoke() three times
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04/gpu_suite.log 2>&1; echo "suite rc $?"; grep -E "passed|failed|error" gpurun_out/r04/gpu_suite.log | tail -3
for i in 1 2 3; do python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke_$i.log 2>&1; echo "smoke $i rc $? $(tail -1 gpurun_out/r04/smoke_$i.log)"; done
