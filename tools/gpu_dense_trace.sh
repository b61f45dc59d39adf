#!/bin/bash
# kernel trace of the dense-seed regime micro run (one slot): per-kernel durations
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/dtrace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dtrace -- python3 tools/dense_micro.py > gpurun_out/dtrace.json 2> gpurun_out/dtrace.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/dtrace/*/*kernel_trace.csv")
rows = list(csv.DictReader(open(f[0])))
tot = collections.defaultdict(int); cnt = collections.Counter()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][:44]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:18]:
    print("  %-46s %6d calls  %9.3f ms total  %9.1f us avg" % (n, cnt[n], v/1e6, v/1e3/cnt[n]))
PY
tail -1 gpurun_out/dtrace.json
