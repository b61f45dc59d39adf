#!/usr/bin/env python3
"""Regenerates tests/golden/*.json from the ORACLE (the reference itself cannot be built here: Go + Plan-9 assembly, no
Go toolchain).  The fixtures freeze the oracle's output on seeded synthetic inputs so that neither the oracle nor the
product can drift unnoticed: tests/test_golden.py checks the oracle against them on CPU and the GPU pipeline against
them on an MI355X.

    python tools/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as O  # noqa: E402

CASES = {
    # name: (seed, genome, reads, read_len, error, variable_length, k, max_rounds, extra OverlapRun kwargs)
    "tiny_k10": (11, 30000, 60, 3000, 0.0, False, 10, -1, {}),
    "tiny_k10_noisy_variable": (13, 40000, 80, 3000, 0.02, True, 10, -1, {}),
    # k=13 needs > 1 % of 4^13 distinct k-mers in the input, or the top-occurrence blacklist leaves no seed (HISTORY.md 2.7)
    "k13_3000x10kb_2rounds": (113, 1500000, 3000, 10000, 0.0, False, 13, 2, {}),
    "tiny_k13_degenerate": (12, 6000, 20, 3000, 0.0, False, 13, -1, {}),
    "short_reads_get_ignored_k10": (32, 60000, 500, 1500, 0.0, True, 10, -1, {}),
    "config1_k10": (1, 250000, 1000, 5000, 0.0, False, 10, -1, {}),
    "query_all_k10": (58, 90000, 350, 4200, 0.01, True, 10, 3, {"query_type": 4}),
}


MAP_CASES = {
    # name: (seed, genome, reads, read_len, error, variable_length, circular, k)
    "map_circular_k11": (3, 200000, 300, 8000, 0.0, False, True, 11),
    "map_noisy_variable_k11": (4, 150000, 300, 6000, 0.05, True, True, 11),
    "map_linear_10pct_k11": (5, 300000, 200, 9000, 0.10, True, False, 11),
}


def build_map(name):
    seed, G, N, L, e, var, circular, k = MAP_CASES[name]
    genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    oref = O.ReadSet(genome, np.array([0, G], dtype=np.int64), min_len=0, himem=False)
    oreads = O.ReadSet(bases, off, min_len=500, himem=False)
    paf, err = O.map_run(oref, oreads, circular=circular, k=k)
    return {"case": name, "generator": {"seed": seed, "genome": G, "reads": N, "read_len": L, "error": e, "variable": var},
            "circular": circular, "k": k, "paf_lines": paf.count("\n"), "paf_sha256": hashlib.sha256(paf.encode()).hexdigest(),
            "paf_head": paf.split("\n")[:8], "stderr": err}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build(name):
    seed, G, N, L, e, var, k, max_rounds, kw = CASES[name]
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    rs = O.ReadSet(bases, off, min_len=1000)
    run = O.OverlapRun(rs, k=k, max_rounds=max_rounds, traces=True, **kw)
    rounds = []
    for r in range(run.rounds):
        paf = run.trace_paf(r)
        cd, co = run.trace(r, "candidates")
        ma, mo = run.trace(r, "matchA")
        mb, _ = run.trace(r, "matchB")
        isg, io = run.trace(r, "indexedSegments")
        rounds.append({
            "seed_kmers_sha256": sha(run.trace(r, "seedKmers").astype(np.int64)), "n_seeds": int(len(run.trace(r, "seedKmers"))),
            "n_queries": int(len(run.trace(r, "queryIDs"))), "n_indexed": int(len(io) - 1),
            "indexed_segments_sha256": sha(isg.astype(np.int64)), "candidates_sha256": sha(cd.astype(np.int64)),
            "n_candidates": int(len(cd)), "n_matches": int(len(mo) - 1),
            "match_a_sha256": sha(ma.astype(np.int64)), "match_b_sha256": sha(mb.astype(np.int64)),
            "paf_lines": paf.count("\n"), "paf_sha256": hashlib.sha256(paf.encode()).hexdigest(),
            "newly_ignored": [int(x) for x in run.trace(r, "newlyIgnored")],
        })
    paf = run.paf
    return {"case": name, "generator": {"seed": seed, "genome": G, "reads": N, "read_len": L, "error": e, "variable": var},
            "k": k, "max_rounds": max_rounds, "kwargs": kw, "rounds": rounds, "n_rounds": run.rounds,
            "paf_lines": paf.count("\n"), "paf_sha256": hashlib.sha256(paf.encode()).hexdigest(),
            "paf_head": paf.split("\n")[:12], "ignored_reads": int(rs.ignore().sum()),
            "input_sha256": sha(np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases)}


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name in CASES:
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(build(name), f, indent=1)
        print("wrote", name)
    for name in MAP_CASES:
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(build_map(name), f, indent=1)
        print("wrote", name)
