#!/bin/bash
# planner lanes: parity tests of the overlap path, the pipeline counters of one job, then an A/B of 1 / 2 / 3 / 4 lanes
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_planner_epoch.py tests/test_gpu_overlap_e2e.py -m gpu -x -q 2>&1 | tail -6
for L in 1 3; do
  DPH_PLAN_LANES=$L DPH_PROFILE=1 timeout 600 python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 > gpurun_out/lanes_$L.json 2> gpurun_out/lanes_$L.err
  echo "lanes $L rc=$?"; grep -E "^\[pipe\]" gpurun_out/lanes_$L.err | tail -8
done
SLOTS=${SLOTS:-6} REPS=${REPS:-3} timeout 1200 python3 tools/ab.py l1:.:DPH_PLAN_LANES=1 l2:.:DPH_PLAN_LANES=2 l3:.:DPH_PLAN_LANES=3 l4:.:DPH_PLAN_LANES=4
