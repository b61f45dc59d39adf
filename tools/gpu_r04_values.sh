#!/bin/bash
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "value or histogram or select" > gpurun_out/r04/values_tests.log 2>&1; grep -E "passed|failed|Error" gpurun_out/r04/values_tests.log | tail -3
DPH_PROFILE=1 timeout 600 python3 bench.py --steps 2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 > gpurun_out/r04/setup_prof2.json 2> gpurun_out/r04/setup_prof2.err; grep "\[setup\]" gpurun_out/r04/setup_prof2.err | tail -6
python3 -c "
import json; d=json.load(open('gpurun_out/r04/setup_prof2.json')); print(d['value']/1e6, d['parity'], d['job_breakdown_s']['setup_value_table_kmer_index_slots'])"
