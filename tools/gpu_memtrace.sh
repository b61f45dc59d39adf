#!/bin/bash
# kernel + memory-copy trace of a short bench run: copy engine operations by direction and size class
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/mtrace
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/mtrace -- python3 bench.py --steps 1 --warmup 0 --max-rounds ${ROUNDS:-150} --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots ${SLOTS:-8} > gpurun_out/mtrace_bench.json 2> gpurun_out/mtrace_bench.err; echo "rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/mtrace/*/*memory_copy_trace.csv")
if not f:
    print("no memory copy trace"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
print("columns:", list(rows[0].keys()) if rows else None, "rows", len(rows))
acc = collections.defaultdict(list)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    b = int(r.get("Bytes", r.get("Size", 0)) or 0)
    cls = "<=4K" if b <= 4096 else "<=64K" if b <= 65536 else "<=1M" if b <= (1 << 20) else ">1M"
    acc[(r.get("Direction", "?"), cls)].append(d)
for k, v in sorted(acc.items()):
    v.sort()
    print("  %-28s %-6s %6d copies  total %9.3f ms  avg %8.1f us  median %8.1f us  max %8.1f us" % (k[0], k[1], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3))
PY
