#!/bin/bash
# Dense-regime (k = 10) index query under the profiler alone: rocprofv3 --kernel-trace --stats of the first rounds of a k = 10 job
# on BASELINE config 2's reads, no legs, no CPU baseline.  SLOTS=1: one round in flight (the kernel's own duration);
# the default slot count otherwise.  Output: gpurun_out/$OUT/dense_s${SLOTS}_kernel_stats.csv + the bench JSON of the same run.
OUT=${OUT:-r03}
mkdir -p gpurun_out/$OUT; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for S in ${SLOTS_LIST:-1 5}; do
  D=gpurun_out/$OUT/dense_s$S
  rm -rf $D
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --k 10 --steps 1 --warmup 0 --max-rounds ${ROUNDS:-12} --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --cpu-rounds 0 --slots $S > $D.json 2> $D.err; echo "dense slots=$S rc=$?"
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/$OUT/dense_s${S}_kernel_stats.csv
  t=$(find $D -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/ktrace_digest.py $t > gpurun_out/$OUT/dense_s${S}_kernel_trace_digest.txt
  rm -rf $D
  grep -i "query_kernel" gpurun_out/$OUT/dense_s${S}_kernel_stats.csv | cut -c1-40,300-
  grep "query_kernel" gpurun_out/$OUT/dense_s${S}_kernel_trace_digest.txt
done
