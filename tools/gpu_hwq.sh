#!/bin/bash
# does the number of HIP hardware queues limit the executor slots' overlap?
for q in 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q SLOTS_LIST="8" bash tools/gpu_slots.sh
done
