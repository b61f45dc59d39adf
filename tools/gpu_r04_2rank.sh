#!/bin/bash
# two ranks sharing the one GPU (gloo exchange), round-parallel layout: the ranks' planners with and without ownership
mkdir -p gpurun_out/r04
for sp in 1 0; do
  DPH_PLAN_SPARSE=$sp DP_BENCH_SAME_DEVICE=1 DP_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus ${RANKS:-2} --steps 2 --warmup 1 --cpu-rounds 0 --mode round --slots ${SLOTS:-3} > gpurun_out/r04/round2_sparse$sp.json 2> gpurun_out/r04/round2_sparse$sp.err; echo "sparse=$sp rc $?"
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04/round2_sparse$sp.json") if l.startswith("{")][-1])
print("value %.2fM ms/job %.1f parity %s" % (d["value"]/1e6, d["ms_per_step"], d["parity"]["paf_sha256_matches_oracle_fixture"]))
for r in d["per_rank"]:
    print("  rank", r["rank"], "setup %.1f ms rounds %.1f ms" % (1e3*r["setup_s"], 1e3*r["rounds_s"]), {k: round(v,1) for k,v in r["per_job"].items()})
PY
done
