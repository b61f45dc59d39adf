#!/bin/bash
# kernel trace of the k = 10 job, one slot and five (per-kernel durations alone and under load)
mkdir -p gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r06
OFF="--cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for s in ${SLOTS:-1 5}; do
  rm -rf $R/kt
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kt -- python3 bench.py --k 10 --steps 1 --warmup 1 --slots $s $OFF > $R/k10_ktrace_${TAG:-x}_slots$s.json 2> $R/k10_ktrace_slots$s.err; echo "ktrace $s rc=$?"
  t=$(find $R/kt -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/ktrace_digest.py $t > $R/k10_kernel_trace_digest_${TAG:-x}_slots$s.txt
  rm -rf $R/kt
  head -24 $R/k10_kernel_trace_digest_${TAG:-x}_slots$s.txt
done
