#!/usr/bin/env python3
"""CPU-only profile of the product's host consensus/PAF stage (finalCheck) on oracle-produced rounds.

Build the instrumented library first (cycle counters per phase, compiled out of the normal build):
    cd downpore_amd/csrc && g++ $(HOSTFLAGS) -DDPH_FINE -shared -o /tmp/fine/libdownpore_host.so host/*.cpp ...
then  DP_HOST_THREADS=1 python tools/host_consensus_profile.py --lib /tmp/fine/libdownpore_host.so
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--reads", type=int, default=3000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--error", type=float, default=0.0)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from downpore_amd import overlap as ov
    if a.lib:
        ov.host_lib_path = lambda: a.lib
    from tests import oracle_lib as O
    from tests.test_host_finalcheck import run_finalcheck
    H = ov.load_host()
    N, L = a.reads, a.read_len
    bases, off = O.gen_reads(113, N * L // 20, N, L, a.error, False)
    rs = O.ReadSet(bases, off, min_len=1000)
    run = O.OverlapRun(rs, k=a.k, max_rounds=a.rounds, traces=True)
    reads = ov.Reads(bases, off, min_len=1000)
    lines = 0
    t0 = time.perf_counter()
    c0 = time.process_time()
    for _ in range(a.reps):
        for r in range(run.rounds):
            paf, _ = run_finalcheck(H, reads, a.k, 1000, run, r)
            lines += paf.count("\n")
    t1 = time.perf_counter()
    print("rounds %d x reps %d: %d PAF lines, %.3f s wall, %.3f s CPU (includes the Python marshalling of the traces)"
          % (run.rounds, a.reps, lines, t1 - t0, time.process_time() - c0))


if __name__ == "__main__":
    main()
