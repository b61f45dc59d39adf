#!/bin/bash
one() { echo -n "$1: "; env $1 SLOTS_LIST=8 bash tools/gpu_slots.sh; }
for i in 1 2 3; do
one DP_X_STAGE=1
one DP_X_STAGE=0
one DP_X_STAGE=2
( cd _ab/prev && echo -n "prev: " && SLOTS_LIST=8 bash tools/gpu_slots.sh )
done
