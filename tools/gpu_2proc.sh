#!/bin/bash
# Is the per-round limit in the process (runtime) or in the GPU?  One process with 6 slots against two / three independent
# processes on the same GPU with 3 / 2 slots each (each runs its own whole jobs; the sum of their rates is what the GPU sustains).
mkdir -p gpurun_out
run() { # slots steps tag
  timeout 600 python3 bench.py --steps $2 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --map-leg-repeats 0 --slots $1 > gpurun_out/p2_$3.json 2> gpurun_out/p2_$3.err
}
show() { python3 - "$@" <<'PY'
import json,sys
tot=0
for t in sys.argv[1:]:
    try:
        d=json.loads([l for l in open("gpurun_out/p2_%s.json"%t) if l.startswith("{")][-1])
        r=1.0/d["rounds_only"]["ms_per_round"]; tot+=r
        print("  %s: %.3f ms/round, job %.3f s, per-job %s, parity %s"%(t,d["rounds_only"]["ms_per_round"],d["job_breakdown_s"]["whole_job"],["%.3f"%x for x in d["job_breakdown_s"]["per_job"]],d["parity"]["paf_sha256_matches_oracle_fixture"]))
    except Exception as e: print("  %s: failed %s"%(t,e))
print("  sum: %.2f rounds/ms -> %.3f ms/round equivalent"%(tot,1/tot if tot else 0))
PY
}
echo "one process, 5 slots"; run 5 6 a; show a
echo "two processes, 3 slots each"; DP_HOST_THREADS=8 run 3 10 b1 & DP_HOST_THREADS=8 run 3 10 b2 & wait; show b1 b2
if [ -n "$ALL" ]; then
echo "three processes, 2 slots each"; DP_HOST_THREADS=6 run 2 10 c1 & DP_HOST_THREADS=6 run 2 10 c2 & DP_HOST_THREADS=6 run 2 10 c3 & wait; show c1 c2 c3
echo "two processes, 6 slots each"; DP_HOST_THREADS=9 run 6 10 d1 & DP_HOST_THREADS=9 run 6 10 d2 & wait; show d1 d2
fi
