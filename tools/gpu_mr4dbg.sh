#!/bin/bash
mkdir -p gpurun_out
export DP_BENCH_BACKEND=gloo DP_BENCH_SAME_DEVICE=1 DPH_DEBUG_PLANNER=1
n=${N:-4}
timeout ${T:-150} python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 24 --warmup 4 --cpu-rounds 0 ${EXTRA} > gpurun_out/bench_mrd.json 2> gpurun_out/bench_mrd.err; echo "rc=$?"
grep "\[planner\]\|\[exec\]" gpurun_out/bench_mrd.err | tail -${LINES_OUT:-60}
cat gpurun_out/bench_mrd.json | cut -c1-300
