#!/bin/bash
# PMC passes (HBM traffic per kernel): separate passes, --kernel-trace only
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcf_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_$c -- python3 bench.py --steps 8 --warmup 4 --cpu-rounds 0 --index-steps 30 > gpurun_out/pmcf_$c.json 2> gpurun_out/pmcf_$c.err; echo "$c rc=$?"
done
ls gpurun_out/pmcf_*/*/ | head
