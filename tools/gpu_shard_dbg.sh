#!/bin/bash
# two ranks sharing the GPU in scan-shard mode over gloo, with the crash handler on
mkdir -p gpurun_out
export DP_BENCH_SAME_DEVICE=1 DP_BENCH_BACKEND=gloo DPH_SEGV_TRACE=1 PYTHONFAULTHANDLER=1
for i in 1 2; do
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 2 --steps 1 --warmup 1 --cpu-rounds 0 --mode scan-shard --slots 4 > gpurun_out/shard_dbg_$i.out 2> gpurun_out/shard_dbg_$i.err
echo "run $i rc=$?"
grep -n "\[dph\] fatal" -A28 gpurun_out/shard_dbg_$i.err | head -80
tail -c 600 gpurun_out/shard_dbg_$i.out
done
