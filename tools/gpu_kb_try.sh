#!/bin/bash
# Durations of the k-mer index build kernels (kb_*) of config-2 jobs + parity of the full-size fixtures.
OUT=${OUT:-r03}
mkdir -p gpurun_out/$OUT; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
D=gpurun_out/$OUT/kb_try
rm -rf $D
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > $D.json 2> $D.err; echo "rc=$?"
s=$(find $D -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && grep '"kb_' $s | cut -c1-40,150-400 > gpurun_out/$OUT/kb_try_stats.txt
[ -n "$s" ] && grep '"kb_' $s | awk -F'",' '{print substr($1,2,24), $2}' 
rm -rf $D
[ -n "$SKIP_TESTS" ] || timeout 900 python3 -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -3
