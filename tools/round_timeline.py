#!/usr/bin/env python3
"""One round's kernels in order, with durations and the gaps between them, from a (gzipped) rocprofv3 kernel_trace.csv of a
one-slot run (tools/gpu_kscale.sh KEEP=1).  Usage: round_timeline.py trace.csv.gz [round_no]"""
import csv, gzip, re, sys
f = sys.argv[1]
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"dp_multi<(.*?), \d+>\(", n)
    return (m.group(1) if m else n.split("(")[0])[:44]
idx = [i for i, r in enumerate(rows) if "kidx_prepare" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
lo, hi = idx[k], idx[k + 1]
prev = None; tot = 0; gaps = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-44s %8.1f us  gap %7.1f  grid %8s lds %6s vgpr %4s" % (short(r["Kernel_Name"]), (e - s) / 1e3, gap, r["Grid_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"]))
    tot += e - s; gaps += max(0, s - prev) if prev else 0; prev = e
print("kernels %.1f us, gaps %.1f us" % (tot / 1e3, gaps / 1e3))
