#!/bin/bash
# two ranks on ONE GPU (test hooks DP_BENCH_SAME_DEVICE / DP_BENCH_BACKEND=gloo): does every multi-GPU mode still run and agree?
mkdir -p gpurun_out
for mode in round scan-shard; do
  DP_BENCH_SAME_DEVICE=1 DP_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --cpu-rounds 0 --mode $mode --slots ${SLOTS:-4} > gpurun_out/mr_$mode.json 2> gpurun_out/mr_$mode.err; echo "$mode rc=$?"
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/mr_$mode.json") if l.startswith("{")][-1])
    print("$mode: value %.2f M/s, job %.3f s, %.3f ms/round, parity %s" % (d["value"]/1e6, d["job_breakdown_s"]["whole_job"], d["rounds_only"]["ms_per_round"], d["parity"]))
except Exception as e:
    print("no result:", e); print(open("gpurun_out/mr_$mode.err").read()[-1500:])
PY
done
