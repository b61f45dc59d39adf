#!/bin/bash
# what does PLACING a mid-sized launch cost a five-slot round?  n extra launches of 300 workgroups x 256 threads that spin 3 us and touch
# no memory behind every count walk (DP_KX_DUMMY=30+n), against n empty launches (10+n), one slot and five
R=gpurun_out/r05; mkdir -p $R
for s in 5 1; do
SLOTS=$s REPS=${REPS:-3} timeout 1500 python3 tools/ab.py base:.: spin4:.:DP_KX_DUMMY=34 spin8:.:DP_KX_DUMMY=38 empty8:.:DP_KX_DUMMY=18 2>&1 | grep -v "committing\|host" | cut -c1-120 | tee $R/ab_placing_s$s.txt
done
