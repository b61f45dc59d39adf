#!/bin/bash
# per-kernel average durations of k = 13 config-2 jobs under five slots: the round-5 build (_ab/prev) and the working tree
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for d in _ab/prev .; do
  rm -rf /tmp/kt_cmp
  (cd $d && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_cmp -- python3 bench.py --steps 3 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --dense-job 0 --map-leg-repeats 0 > /tmp/kt_cmp.json 2>/dev/null)
  f=$(find /tmp/kt_cmp -name "*kernel_stats.csv" | head -1)
  echo "== $d"
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = n.split("(")[0].replace("void ", "")
    n = re.sub(r"^dp_kernel<(.*)>\s*$", r"\1", n.strip())
    n = re.sub(r"^dp_multi<(.*), 1>\s*$", r"\1", n.strip())
    return n.strip().rstrip(">").strip() if n.startswith("dp_") else n.strip()
out = []
for r in rows:
    n = short(r["Name"])
    if n.startswith("kb_") or "rocclr" in n or n.startswith("values_") or "rocprim" in n or n in ("pack_kernel", "select_kernel", "hist_kernel"): continue
    out.append((float(r["TotalDurationNs"]), n, int(r["Calls"]), float(r["AverageNs"])))
for tot, n, c, avg in sorted(out, reverse=True)[:22]:
    print("%-44s calls %6d  avg %7.1f us  total %8.1f ms" % (n[:44], c, avg / 1e3, tot / 1e6))
PY
done
