#!/bin/bash
# what n empty launches per round cost: one slot against five
mkdir -p gpurun_out/r04
for S in 1 5; do
  SLOTS=$S REPS=3 NAME=empty_s$S VARIANTS="none:.: e10:.:DP_KX_DUMMY=20 e20:.:DP_KX_DUMMY=30" bash tools/gpu_r04_ab.sh
done
