#!/bin/bash
# round 4: (1) the fixed failure test, (2) the dense regime's runtime copies by call site (COPYLOG build) and by kernel trace,
# (3) a one-slot round timeline of the sparse regime for reference
mkdir -p gpurun_out/r04; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "rank_failure" > gpurun_out/r04/fail_tests.log 2>&1; echo "fail tests rc $?"; tail -2 gpurun_out/r04/fail_tests.log
DP_LIB_DIR=$PWD/downpore_amd/lib_copylog timeout 600 python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds 12 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots 1 > gpurun_out/r04/dense_copylog.json 2> gpurun_out/r04/dense_copylog.err; echo "copylog rc $?"
grep copylog gpurun_out/r04/dense_copylog.err | sort -t' ' -k7 -n -r | head -40
D=gpurun_out/r04/dense_trace
rm -rf $D
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -- python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds 12 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots 1 > $D.json 2> $D.err; echo "trace rc $?"
f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04/dense_s1_kernel_stats.csv
t=$(find $D -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/ktrace_digest.py $t > gpurun_out/r04/dense_s1_kernel_trace_digest.txt
m=$(find $D -name "*memory_copy_trace.csv" | head -1); [ -n "$m" ] && cp $m gpurun_out/r04/dense_s1_memory_copy_trace.csv
[ -n "$t" ] && grep -i copyBuffer $t | head -400 > gpurun_out/r04/dense_s1_copybuffer_rows.csv
[ -n "$t" ] && head -1 $t > gpurun_out/r04/dense_s1_trace_header.csv
rm -rf $D
head -25 gpurun_out/r04/dense_s1_kernel_stats.csv | cut -c1-150
