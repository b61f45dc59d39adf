#!/usr/bin/env python3
"""Which kernels slow which?  From a rocprofv3 kernel_trace.csv of a run with several executor slots: for every launch of kernel
type A inside the stretches of rounds, the fraction of its duration during which a launch of type B (another stream) was running;
then per A a least-squares fit  duration = base + sum_B coef_B * frac_B.  coef_B / base = relative slow-down of A while B runs.
Prints, per A, the base (what it takes alone, by the fit), its mean, and the B's with the largest contribution to the mean."""
import collections, csv, re, sys
import numpy as np

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    m = re.search(r"dp_multi<(.*?), \d+>\(", n)
    n = (m.group(1) if m else n.split("(")[0])[:40]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
# stretches of rounds: cut at idle gaps >= 2 ms, keep stretches with >= 100 chain_walk launches
segs = []; cs = ce = None; ncw = 0
for s, e, n, q in rows:
    if ce is None or s - ce >= 2000000:
        if ce is not None: segs.append((cs, ce, ncw))
        cs, ce, ncw = s, e, 0
    ce = max(ce, e); ncw += 1 if "chain_walk" in n else 0
segs.append((cs, ce, ncw))
segs = [(a, b) for a, b, c in segs if c >= 100]
inr = [x for x in rows if any(a <= x[0] and x[1] <= b for a, b in segs)]
types = sorted(set(x[2] for x in inr))
tix = {t: i for i, t in enumerate(types)}
# sweep: for each launch, overlap with every other launch (different queue); launches sorted by start
starts = np.array([x[0] for x in inr]); ends = np.array([x[1] for x in inr])
X = collections.defaultdict(list); Y = collections.defaultdict(list)
j0 = 0
for i, (s, e, n, q) in enumerate(inr):
    d = e - s
    if d <= 0: continue
    f = np.zeros(len(types))
    # candidates: launches that start before e and end after s
    lo = np.searchsorted(starts, s - 2000000)  # (no kernel of a round is longer than 2 ms)
    hi = np.searchsorted(starts, e)
    for j in range(lo, hi):
        if j == i: continue
        s2, e2, n2, q2 = inr[j]
        ov = min(e, e2) - max(s, s2)
        if ov > 0: f[tix[n2]] += ov / d
    X[n].append(f); Y[n].append(d / 1e3)
print("%-34s %6s %8s %8s   largest contributions to the mean (kernel: us added = coef x mean overlap fraction; coef/base)" % ("kernel", "n", "mean_us", "base_us"))
for n in sorted(X, key=lambda k: -sum(Y[k])):
    A = np.array(X[n]); y = np.array(Y[n])
    if len(y) < 200: continue
    keep = A.mean(axis=0) > 0.02
    A2 = np.hstack([np.ones((len(y), 1)), A[:, keep]])
    coef, *_ = np.linalg.lstsq(A2, y, rcond=None)
    base = coef[0]; names = [t for t, k in zip(types, keep) if k]
    contrib = sorted(((c * A[:, keep][:, i].mean(), c / max(base, 1e-9), nm) for i, (c, nm) in enumerate(zip(coef[1:], names))), reverse=True)
    print("%-34s %6d %8.1f %8.1f   %s" % (n, len(y), y.mean(), base, "; ".join("%s: %+.1f (%.2f)" % (nm[:28], c, rel) for c, rel, nm in contrib[:5])))
