#!/bin/bash
R=gpurun_out/r05; mkdir -p $R
timeout 900 python -m pytest tests/test_gpu_overlap_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "counting_step or kmer_index or index_mode or sort_tiers or scan_prepare or compaction or shares" > $R/fuse_tests.log 2>&1; echo "tests rc $?"; tail -2 $R/fuse_tests.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "config2_whole_job_matches or flagged or config4_first or dense" > $R/fuse_full.log 2>&1; echo "full-size rc $?"; tail -2 $R/fuse_full.log
REPS=${REPS:-4} timeout 1500 python3 tools/ab.py two:.:DP_KX_FUSE=0 count:.:DP_KX_FUSE=c both:.:DP_KX_FUSE=1 2>&1 | grep -v "committing\|host:" | cut -c1-130 | tee $R/ab_fuse_count_into_offsets.txt
