#!/bin/bash
# Per-kernel durations of a round alone (one slot) and of four rounds sharing every launch (one gang of four): how each kernel
# scales when it carries more rounds.  rocprofv3 --kernel-trace --stats of the first ROUNDS rounds of a config-2 job.
OUT=${OUT:-r03}
mkdir -p gpurun_out/$OUT; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in ${CFGS:-1:1 4:4}; do
  G=${cfg%%:*}; S=${cfg##*:}
  D=gpurun_out/$OUT/kscale_g${G}_s${S}
  rm -rf $D
  DPH_GANG=$G timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 1 --warmup 0 --max-rounds ${ROUNDS:-160} --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots $S ${BENCH_ARGS} > $D.json 2> $D.err; echo "g=$G s=$S rc=$?"
  t=$(find $D -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/ktrace_digest.py $t > gpurun_out/$OUT/kscale_g${G}_s${S}_digest.txt
  [ -n "$KEEP" ] && [ -n "$t" ] && gzip -c $t > gpurun_out/$OUT/kscale_g${G}_s${S}_trace.csv.gz
  rm -rf $D
  head -24 gpurun_out/$OUT/kscale_g${G}_s${S}_digest.txt | cut -c1-130
done
