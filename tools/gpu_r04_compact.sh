#!/bin/bash
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_overlap_e2e.py tests/test_gpu_kernels.py -x -q -m gpu -k "kmer_index or paf_bit_exact or chunks_made or kindex or value" 2>&1 | tail -5
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > gpurun_out/r04/setup_narrow.json 2> gpurun_out/r04/setup_narrow.err; grep "\[setup\]" gpurun_out/r04/setup_narrow.err | tail -6
python3 -c "
import json; d=json.load(open('gpurun_out/r04/setup_narrow.json')); print(d['value']/1e6, d['parity'], d['job_breakdown_s'])"
