#!/bin/bash
# generic A/B on one box: VARIANTS="label:dir:ENV=V,... label2:dir:..." REPS=n
mkdir -p gpurun_out/r04
REPS=${REPS:-3} timeout 1500 python3 tools/ab.py $VARIANTS 2>&1 | tee gpurun_out/r04/ab_${NAME:-last}.txt
