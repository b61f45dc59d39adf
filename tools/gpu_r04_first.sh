#!/bin/bash
# round 4, first GPU call: smoke three times (the round-3 failure was a race), the new tests, a driver-style bench line
mkdir -p gpurun_out/r04
for i in 1 2 3; do python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/smoke_$i.log 2>&1; echo "smoke $i rc $?"; done
python -m pytest tests/test_gpu_bench.py tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "plain_launch or rank_failure or gang" > gpurun_out/r04/new_tests.log 2>&1; echo "new tests rc $?"; tail -3 gpurun_out/r04/new_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_start.json 2> gpurun_out/r04/bench_start.err; echo "bench rc $?"
cat gpurun_out/r04/bench_start.json | cut -c1-600
