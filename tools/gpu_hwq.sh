#!/bin/bash
# executor slots x HIP hardware queues
for q in 8 16; do
  for s in 8 12 16; do
    echo -n "GPU_MAX_HW_QUEUES=$q "; GPU_MAX_HW_QUEUES=$q SLOTS_LIST="$s" bash tools/gpu_slots.sh
  done
done
