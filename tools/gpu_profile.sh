#!/bin/bash
# final-configuration bench + rocprofv3 kernel stats of the same command (+ PMC passes)
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_final; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --cpu-rounds 0 > gpurun_out/bench_final_prof.json 2> gpurun_out/bench_final_prof.err; echo "prof rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcf_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_$c -- python3 bench.py --steps 8 --warmup 4 --cpu-rounds 0 > gpurun_out/pmcf_$c.json 2> gpurun_out/pmcf_$c.err; echo "$c rc=$?"
done
cat gpurun_out/bench_final.json | head -c 3000
