#!/bin/bash
# one slot, first 120 rounds under the kernel trace: the kernels of one round in order with their gaps (final code of the round)
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
C="--steps 1 --warmup 0 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 --slots 1"
rm -rf $R/kt1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/kt1 -- python3 bench.py $C --max-rounds 120 > $R/kt1.json 2> $R/kt1.err; echo "trace rc=$?"
t=$(find $R/kt1 -name "*kernel_trace.csv" | head -1)
python3 tools/round_timeline.py $t 60 > $R/round_timeline_one_slot_final.txt
for r in 30 50 70 90; do python3 tools/round_timeline.py $t $r | grep -E "kidx_|kernels "; done > $R/round_timeline_one_slot_final_kidx_of_four_rounds.txt
cat $R/round_timeline_one_slot_final.txt
rm -rf $R/kt1 $R/kt1.json $R/kt1.err
