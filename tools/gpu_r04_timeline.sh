#!/bin/bash
# one slot, first ROUNDS rounds under the kernel trace: the kernels of one round in order with their gaps (tools/round_timeline.py)
mkdir -p gpurun_out/r04; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
R=gpurun_out/r04
rm -rf $R/kt1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py --steps 1 --warmup 0 --max-rounds ${ROUNDS:-120} --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots 1 > $R/kt1.json 2> $R/kt1.err; echo "rc=$?"
t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
python3 tools/round_timeline.py $t > $R/round_timeline_one_slot${TAG}.txt
for r in 20 40 80 100; do python3 tools/round_timeline.py $t $r | grep -E "kidx_prepare|kernels "; done
rm -rf $R/kt1
cat $R/round_timeline_one_slot${TAG}.txt
