#!/usr/bin/env python3
"""Per-kernel digest of a rocprofv3 kernel_trace.csv that is too large to carry back from the GPU box: calls, total, mean,
median, max, and the GPU-busy fraction (union of the kernel intervals) between the first and the last chain_walk_kernel."""
import collections, csv, sys
tot = collections.defaultdict(int); seq = collections.defaultdict(list); ev = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void dp_multi<", "<").split("(")[0][:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot[n] += e - s; seq[n].append(e - s); ev.append((s, e, "chain_walk_kernel" in n))
print("%-62s %8s %12s %10s %10s %10s" % ("kernel", "calls", "total_ms", "mean_us", "median_us", "max_us"))
for n, v in sorted(tot.items(), key=lambda x: -x[1]):
    s = sorted(seq[n])
    print("%-62s %8d %12.3f %10.1f %10.1f %10.1f" % (n, len(s), v / 1e6, v / 1e3 / len(s), s[len(s) // 2] / 1e3, s[-1] / 1e3))
ev.sort()
cw = [(s, e) for s, e, c in ev if c]
if cw:
    t0, t1 = cw[0][0], max(e for _, e in cw)
    busy = ksum = 0; cs = ce = None
    for s, e, _ in ev:
        if e < t0 or s > t1: continue
        ksum += e - s
        if ce is None or s > ce:
            if ce is not None: busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("window first..last chain_walk_kernel: %.1f ms, GPU busy (union of kernel intervals) %.1f ms = %.0f %%, sum of kernel times %.1f ms"
          % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), ksum / 1e6))
