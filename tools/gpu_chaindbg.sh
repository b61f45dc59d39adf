#!/bin/bash
mkdir -p gpurun_out
DP_CHAIN_DEBUG=1 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-rounds 0 --slots 1 > gpurun_out/bench_cd.json 2> gpurun_out/bench_cd.err
grep "\[chain\]" gpurun_out/bench_cd.err | tail -6
