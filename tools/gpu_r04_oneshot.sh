#!/bin/bash
# round 4: the index step in one go (hit records, no wait between count and write) - parity tests, then A/B against the two-step form
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_overlap_e2e.py -x -q -m gpu > gpurun_out/r04/oneshot_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r04/oneshot_tests.log
python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r04/oneshot_full.log 2>&1; echo "full-size rc $?"; tail -3 gpurun_out/r04/oneshot_full.log
REPS=3 python3 tools/ab.py two:.:DP_KX_ONESHOT=0 one:.:DP_KX_ONESHOT=1,DP_KX_ONESHOT_DEBUG=1 2>&1 | tee gpurun_out/r04/ab_oneshot.txt
DP_KX_ONESHOT_DEBUG=1 python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 6 --map-leg-repeats 0 2> gpurun_out/r04/oneshot_dbg.err > gpurun_out/r04/oneshot_dbg.json; grep -c "one-go step repeated" gpurun_out/r04/oneshot_dbg.err; grep "one-go step repeated" gpurun_out/r04/oneshot_dbg.err | tail -3
