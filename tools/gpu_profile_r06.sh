#!/bin/bash
# Round-6 evidence, run on the GPU box (gpurun -- bash tools/gpu_profile_r06.sh):
#  1. rocprofv3 --kernel-trace --stats of the default bench command          -> gpurun_out/r06/ktrace
#  2. PMC passes (FETCH_SIZE | WRITE_SIZE | SQ set; one pass each, --kernel-trace only) over three short workloads:
#       main  = first 40 rounds of a config-2 job, resident k-mer position index (set-up kernels at full size)
#       scan  = the same with the scan kernels (DP_SCAN_INDEX=0)
#       dense = k=10 (dense seeds): first 6 rounds
# tools/pmc_summary.py turns the CSVs into profiles/r06/*.json
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r06
rm -rf $R/ktrace
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/ktrace -- python3 bench.py > $R/ktrace_bench.json 2> $R/ktrace_bench.err; echo "ktrace rc=$?"
t=$(find $R/ktrace -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/ktrace_digest.py $t > $R/bench_default_kernel_trace_digest.txt
f=$(find $R/ktrace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/bench_default_kernel_stats.csv
cp $R/ktrace_bench.json $R/bench_default_under_rocprof.json
rm -rf $R/ktrace
if [ -n "$KTRACE_ONLY" ]; then du -sh $R; ls $R; exit 0; fi
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
COMMON="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
run() { # name, counters, bench args...
  local name=$1 ctr=$2; shift 2
  rm -rf $R/pmc_$name
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/pmc_$name -- python3 bench.py $COMMON "$@" > $R/pmc_$name.json 2> $R/pmc_$name.err; echo "$name rc=$?"
}
for c in FETCH_SIZE WRITE_SIZE; do
  run main_$c $c --max-rounds 40
  DP_SCAN_INDEX=0 run scan_$c $c --max-rounds 40
  run dense_$c $c --k 10 --max-rounds 6
done
run main_SQ "$SQ" --max-rounds 40
DP_SCAN_INDEX=0 run scan_SQ "$SQ" --max-rounds 40
run dense_SQ "$SQ" --k 10 --max-rounds 6
# the per-dispatch kernel traces of the PMC runs are not needed (the counter CSV names the kernel of every dispatch)
find $R -name "*kernel_trace.csv" -path "*pmc_*" -delete
du -sh $R; ls $R
