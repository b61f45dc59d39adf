#!/bin/bash
# does the consensus kernel's LENGTH cost five slots anything?  every window sleeps 30 us more
for S in 1 5; do
  SLOTS=$S REPS=3 NAME=consspin_s$S VARIANTS="none:.: spin30:.:DP_CONS_SPIN=30" bash tools/gpu_r04_ab.sh
done
