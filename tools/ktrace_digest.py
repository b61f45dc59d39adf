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

# concurrency histogram over the same window: how long exactly n kernels were running at once (n = 0: nothing on the GPU)
if cw:
    pts = []
    for s_, e_, _ in ev:
        if e_ < t0 or s_ > t1: continue
        pts.append((max(s_, t0), 1)); pts.append((min(e_, t1), -1))
    pts.sort()
    hist = collections.defaultdict(int); cur = 0; last = t0
    for t_, d_ in pts:
        hist[min(cur, 8)] += t_ - last; last = t_; cur += d_
    hist[min(cur, 8)] += t1 - last
    tot_ = float(t1 - t0)
    print("kernels running at once (share of the window): " + "  ".join("%d%s: %.1f %%" % (n, "+" if n == 8 else "", 100.0 * hist[n] / tot_) for n in sorted(hist)))

# the same inside the stretches of rounds only: the timeline is cut wherever nothing ran for 2 ms or longer (between jobs the host
# hashes / resets; a job's set-up kernels are long single kernels) and only stretches with >= 100 chain_walk launches count
if cw:
    evs = sorted((s_, e_, c_) for s_, e_, c_ in ev)
    segs = []; cs = ce = None; ncw = 0
    for s_, e_, c_ in evs:
        if ce is None or s_ - ce >= 2000000:
            if ce is not None: segs.append((cs, ce, ncw))
            cs, ce, ncw = s_, e_, 0
        ce = max(ce, e_); ncw += 1 if c_ else 0
    segs.append((cs, ce, ncw))
    segs = [x for x in segs if x[2] >= 100]
    tot_len = sum(e_ - s_ for s_, e_, _ in segs)
    hist = collections.defaultdict(int); ksum2 = 0
    for a_, b_, _ in segs:
        pts = []
        for s_, e_, _c in evs:
            if e_ <= a_ or s_ >= b_: continue
            pts.append((max(s_, a_), 1)); pts.append((min(e_, b_), -1)); ksum2 += min(e_, b_) - max(s_, a_)
        pts.sort(); cur = 0; last = a_
        for t_, d_ in pts:
            hist[min(cur, 8)] += t_ - last; last = t_; cur += d_
        hist[min(cur, 8)] += b_ - last
    if tot_len:
        print("stretches of rounds: %d, %.1f ms in all, %d chain_walk launches; kernel-time sum %.1f ms; kernels running at once: %s" %
              (len(segs), tot_len / 1e6, sum(x[2] for x in segs), ksum2 / 1e6,
               "  ".join("%d%s: %.1f %%" % (n, "+" if n == 8 else "", 100.0 * hist[n] / tot_len) for n in sorted(hist))))
