#!/bin/bash
# phase profile of the host consensus stage on the GPU box's CPUs (instrumented build, CPU only)
mkdir -p gpurun_out/fine
cd downpore_amd/csrc && g++ -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -I../../include -Wall -Wno-unused-parameter -DDPH_FINE -shared -o ../../gpurun_out/fine/libdownpore_host.so host/host_seq.cpp host/host_overlap.cpp host/host_pool.cpp host/host_pipeline.cpp host/host_map.cpp host/host_capi.cpp -L../lib -ldownpore_hip -Wl,-rpath,$PWD/../lib 2>&1 | grep error
cd ../..
for t in 1 16; do
  echo "threads $t"
  DP_HOST_THREADS=$t python tools/host_consensus_profile.py --lib gpurun_out/fine/libdownpore_host.so --reps 40 2>&1 | tail -2
done
