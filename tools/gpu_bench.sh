#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?" >> gpurun_out/bench1.err
rm -rf gpurun_out/prof; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --cpu-rounds 0 > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err; echo "prof rc=$?" >> gpurun_out/bench_prof.err
find gpurun_out/prof -name "*stats*" | head; 
cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench1.json; tail -3 gpurun_out/bench1.err
