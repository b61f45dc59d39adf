#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): do five and more slots share queues?
for S in 5 8; do
  SLOTS=$S REPS=3 NAME=hwq_s$S VARIANTS="q4:.: q8:.:GPU_MAX_HW_QUEUES=8 q16:.:GPU_MAX_HW_QUEUES=16" bash tools/gpu_r04_ab.sh
done
