#!/bin/bash
# index-build kernels: parity of the index tests, then set-up and kernel times
R=gpurun_out/r05; mkdir -p $R
timeout 1200 python -m pytest tests/test_gpu_overlap_e2e.py -x -q -m gpu -k "kmer_index or index_mode or shares" > $R/kb_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/kb_tests.log
bash tools/r05/gpu_kb_time.sh
