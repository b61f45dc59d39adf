#!/bin/bash
# the first milliseconds of a k = 13 job: when rounds get committed, and what the pipeline's counters say, for 10 jobs of one process
python3 - <<'PY'
import os, sys, time, json, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.synth import gen_reads_truth
from downpore_amd.overlap import OverlapPipeline, Reads
bases, off, _, _ = gen_reads_truth(1, 50000000, 100000, 10000, 0.0, False)
reads = Reads(bases, off, min_len=1000, himem=True)
pipe = OverlapPipeline(reads, k=13, seed_batch_size=10000, slots=5, defer_init=True)
H = pipe.H
H.dph_planner_counter.restype = ctypes.c_int64
names = ["plans_computed", "plans_thrown_away", "plans_erased_by_flags", "rounds_executed", "rounds_rejected", "rounds_committed",
         "plan_compute_us", "slot_wait_for_plan_us", "commit_thread_wait_us", "commit_text_us", "commit_state_us", "commit_keep_text_us", "formatter_busy_us", "commit_wait_for_formatter_us"]
def ctr(): return [int(H.dph_planner_counter(i)) for i in range(len(names))]
for j in range(10):
    c0 = ctr()
    t0 = time.perf_counter(); pipe.init(); t1 = time.perf_counter()
    tl = []; rounds = 0
    while True:
        c = pipe.step(); t = time.perf_counter()
        if c == 0: break
        rounds += c; tl.append((t - t1, rounds, c))
    c1 = ctr()
    pipe.reset()
    def at(ms):
        r = 0
        for t, rr, c in tl:
            if t <= ms * 1e-3: r = rr
        return r
    d = {n: c1[i] - c0[i] for i, n in enumerate(names)}
    print("job %d: init %.1f ms, rounds part %.1f ms | committed by 2/5/10/20/40 ms: %d %d %d %d %d | thrown %d erased %d rejected %d executed %d | slot wait for plan %.1f ms, commit idle %.1f ms" %
          (j, 1e3 * (t1 - t0), 1e3 * tl[-1][0], at(2), at(5), at(10), at(20), at(40), d["plans_thrown_away"], d["plans_erased_by_flags"], d["rounds_rejected"], d["rounds_executed"],
           d["slot_wait_for_plan_us"] / 1e3, d["commit_thread_wait_us"] / 1e3), flush=True)
pipe.close()
PY
