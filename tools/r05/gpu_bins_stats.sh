#!/bin/bash
# per-kernel means of whole jobs with five slots in flight, records against bins (rocprofv3 --kernel-trace --stats)
R=gpurun_out/r05; mkdir -p $R; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
C="--steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0"
for v in "DP_KX_BINS=0" "DP_KX_BIN_WAVES=8" "DP_KX_BIN_WAVES=4"; do
  rm -rf $R/kts; export $v
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/kts -- python3 bench.py $C > $R/kts.json 2> $R/kts.err; echo "$v rc=$?"
  unset DP_KX_BINS DP_KX_BIN_WAVES
  f=$(find $R/kts -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r"dp_multi<(.*?), \d+>\(", n)
    return (m.group(1) if m else n.split("(")[0])[:40]
tot = 0
out = []
for r in rows:
    nm = short(r["Name"])
    if int(r["Calls"]) < 1000: continue
    out.append((nm, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    tot += float(r["TotalDurationNs"]) / 1e6
print(sys.argv[2], "sum of per-round kernels %.1f ms" % tot)
for o in sorted(out, key=lambda x: -x[2]): print("   %-42s calls %6d total %8.1f ms mean %7.1f us" % o)
PY
  python3 - <<PY
import json
d=json.loads([l for l in open('$R/kts.json') if l.startswith('{')][-1])
print('   under the trace: ms/round %.4f job %.4f' % (d['rounds_only']['ms_per_round'], d['job_breakdown_s']['whole_job']))
PY
done
rm -rf $R/kts $R/kts.json $R/kts.err
