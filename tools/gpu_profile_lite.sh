#!/bin/bash
# final-configuration bench + rocprofv3 kernel stats of the same command (PMC passes: tools/gpu_profile.sh)
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_final; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --cpu-rounds 0 > gpurun_out/bench_final_prof.json 2> gpurun_out/bench_final_prof.err; echo "prof rc=$?"
find gpurun_out/prof_final -name "*kernel_stats.csv" | head -2
head -c 1500 gpurun_out/bench_final.json
