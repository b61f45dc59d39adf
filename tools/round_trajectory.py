#!/usr/bin/env python3
"""Wall time per committed round over the first rounds of a config-2 job (where do the early rounds lose time?)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.synth import gen_reads
from downpore_amd.overlap import OverlapPipeline, Reads
N, L = 100000, 10000
bases, off = gen_reads(2, N * L // 20, N, L, 0.0, False)
reads = Reads(bases, off, min_len=1000)
pipe = OverlapPipeline(reads, k=13, slots=int(os.environ.get("SLOTS", "6")))
t = [time.perf_counter()]
rounds = [0]
while rounds[-1] < 400:
    c = pipe.step()
    if c == 0:
        break
    rounds.append(rounds[-1] + c)
    t.append(time.perf_counter())
t = np.array(t); r = np.array(rounds)
for lo in range(0, 400, 25):
    sel = (r >= lo) & (r <= lo + 25)
    if sel.sum() > 1:
        tt, rr = t[sel], r[sel]
        print("rounds %3d-%3d: %.3f ms/round" % (lo, lo + 25, 1e3 * (tt[-1] - tt[0]) / max(1, rr[-1] - rr[0])))
