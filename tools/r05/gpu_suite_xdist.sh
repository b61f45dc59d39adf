#!/bin/bash
# the whole -m gpu suite on three pytest-xdist workers sharing the GPU (what is left of the round's GPU minutes does not hold the serial suite)
R=gpurun_out/r05; mkdir -p $R
timeout 440 python -m pytest tests -q -m gpu -n 3 -p no:cacheprovider > $R/gpu_suite_xdist.log 2>&1; echo "suite rc $?"
tail -4 $R/gpu_suite_xdist.log | cut -c1-300
