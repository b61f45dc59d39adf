#!/bin/bash
# kernel trace of the main workload alone (no legs): GPU-busy union and concurrency histogram of the rounds
mkdir -p gpurun_out/r04; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r04
rm -rf $R/ktrace_main
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/ktrace_main -- python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots ${SLOTS:-5} > $R/ktrace_main_bench.json 2> $R/ktrace_main_bench.err; echo "rc=$?"
t=$(find $R/ktrace_main -name "*kernel_trace.csv" | head -1)
python3 tools/ktrace_digest.py $t > $R/main_only_s${SLOTS:-5}_kernel_trace_digest.txt
rm -rf $R/ktrace_main
tail -3 $R/main_only_s${SLOTS:-5}_kernel_trace_digest.txt
