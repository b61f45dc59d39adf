#!/usr/bin/env python3
"""Whole-job parity at the full BASELINE config-2 size: every round of the GPU pipeline against every round of the
oracle (its per-read scans spread over the host cores; everything else sequential).  Writes a JSON summary.

    python tools/full_parity.py [--reads 100000] [--read-len 10000] [--k 13] [--out profiles/r01/full_job_parity.json]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--k", type=int, default=13)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--error", type=float, default=0.0)
    ap.add_argument("--max-rounds", type=int, default=-1)
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r01", "full_job_parity.json"))
    a = ap.parse_args()
    from bench import cpu_budget
    from downpore_amd.overlap import OverlapPipeline, Reads
    from tests import oracle_lib as O
    from tools.synth import gen_reads
    N, L = a.reads, a.read_len
    bases, off = gen_reads(a.seed, N * L // 20, N, L, a.error, False)
    reads = Reads(bases, off, min_len=1000)
    pipe = OverlapPipeline(reads, k=a.k, slots=a.slots)
    t0 = time.time()
    rounds = pipe.run(a.max_rounds)
    t_gpu = time.time() - t0
    gpu_paf = pipe.all_paf()
    values = pipe.values().copy()
    gpu_ignore = reads.ignore().copy()
    last = pipe.stats()
    pipe.close()
    O.build_oracle()
    os.environ["DPO_SCAN_THREADS"] = str(a.threads or cpu_budget())
    rs = O.ReadSet(bases, off, min_len=1000)
    t0 = time.time()
    run = O.OverlapRun(rs, k=a.k, values=np.ascontiguousarray(values), traces=False, max_rounds=rounds if a.max_rounds >= 0 else -1)
    t_cpu = time.time() - t0
    same = gpu_paf == run.paf
    out = {"workload": "%d reads x %d bp, error %.3g, k=%d, %s" % (N, L, a.error, a.k, "every round" if a.max_rounds < 0 else "first %d rounds" % rounds), "rounds_gpu": rounds, "rounds_oracle": run.rounds,
           "paf_lines": gpu_paf.count("\n"), "paf_sha256_gpu": hashlib.sha256(gpu_paf.encode()).hexdigest(),
           "paf_sha256_oracle": hashlib.sha256(run.paf.encode()).hexdigest(), "paf_identical": bool(same),
           "ignore_flags_identical": bool(np.array_equal(gpu_ignore, rs.ignore())),
           "last_round_served_by_kmer_index": bool(last.get("idx_rounds", 0)),
           "total_bases": int(off[-1]), "gpu_pipeline_s": t_gpu, "oracle_s": t_cpu, "oracle_scan_threads": int(os.environ["DPO_SCAN_THREADS"])}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))
    if not same:
        la, lb = gpu_paf.split("\n"), run.paf.split("\n")
        for i, (x, y) in enumerate(zip(la, lb)):
            if x != y:
                print("first difference at line", i, "\n gpu   ", x, "\n oracle", y)
                break
        sys.exit(1)


if __name__ == "__main__":
    main()
