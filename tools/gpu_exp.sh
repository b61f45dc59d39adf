#!/bin/bash
# after the op merges: parity + throughput, with and without timing events
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== 8 slots"; SLOTS_LIST=8 bash tools/gpu_slots.sh
echo "== 8 slots, no timing events"; DP_KERNEL_TIMING=0 SLOTS_LIST=8 bash tools/gpu_slots.sh
echo "== 8 slots, Q=4"; GPU_MAX_HW_QUEUES=4 SLOTS_LIST=8 bash tools/gpu_slots.sh
echo "== 6 slots Q=8"; SLOTS_LIST=6 bash tools/gpu_slots.sh
echo "== 8 slots again"; SLOTS_LIST=8 bash tools/gpu_slots.sh
echo "== 8 slots, no timing events"; DP_KERNEL_TIMING=0 SLOTS_LIST=8 bash tools/gpu_slots.sh
