#!/bin/bash
mkdir -p gpurun_out/r04; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r04
rm -rf $R/kti
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/kti -- python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots ${SLOTS:-5} > $R/kti.json 2> $R/kti.err; echo "rc=$?"
t=$(find $R/kti -name "*kernel_trace.csv" | head -1)
head -1 $t
timeout 900 python3 tools/kt_interference.py $t > $R/interference_s${SLOTS:-5}.txt 2>&1
rm -rf $R/kti
cat $R/interference_s${SLOTS:-5}.txt
