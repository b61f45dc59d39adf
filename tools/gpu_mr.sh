#!/bin/bash
# N ranks on ONE GPU (test hooks DP_BENCH_SAME_DEVICE / DP_BENCH_BACKEND=gloo): do the multi-GPU modes run and agree, and what
# does a second process on the same GPU add (separate HIP runtimes, same device)?
mkdir -p gpurun_out
IFS=";" read -ra LIST <<< "${CFGS:-2 round 4;2 scan-shard 4}"
for cfg in "${LIST[@]}"; do
  IFS=" " read -r n mode slots <<< "$cfg"
  DP_BENCH_SAME_DEVICE=1 DP_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 2 --warmup 1 --cpu-rounds 0 --mode $mode --slots $slots > gpurun_out/mr.json 2> gpurun_out/mr.err; rc=$?
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/mr.json") if l.startswith("{")][-1])
    print("$n ranks x $slots slots, $mode: value %.2f M/s, job %.3f s, %.3f ms/round, parity %s" % (d["value"]/1e6, d["job_breakdown_s"]["whole_job"], d["rounds_only"]["ms_per_round"], d["parity"]["paf_sha256_matches_oracle_fixture"]))
except Exception as e:
    print("$n ranks x $slots slots, $mode: no result (rc $rc):", e); print(open("gpurun_out/mr.err").read()[-800:])
PY
done
