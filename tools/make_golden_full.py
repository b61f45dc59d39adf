#!/usr/bin/env python3
"""Full-size golden fixtures from the ORACLE alone (no GPU, nothing of the product involved): the oracle computes its
own k-mer value table and runs the overlap command on the seeded synthetic set of a BASELINE config; the fixture keeps
the SHA-256 of the PAF text, of the ignore flags and the cumulative line count after every round, so a GPU run at the
same size (tests/test_gpu_full_size.py) can be held to it without the oracle having to run for minutes on the GPU box.

    python tools/make_golden_full.py config2            # 100k x 10 kb, every round (about 20 min on 6 threads)
    python tools/make_golden_full.py config4 --rounds 6 # first rounds of the 1M x 10 kb set
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
from tests import oracle_lib as O  # noqa: E402

CONFIGS = {
    # name: (seed, reads, read_len, error, k, variable read lengths)
    "config2": (2, 100000, 10000, 0.0, 13, False),
    "config4": (4, 1000000, 10000, 0.0, 13, False),
    # SURVEY 8(d)'s other inputs at config 2's size: the dense-seed regime (k = 10, the command's default; the regime in which
    # the index query streams hundreds of MB per round), k = 13 with 0.2 % errors, and the L*U[0.5,1.5] length model - the
    # last two flag reads (SetIgnore, commands/overlap.go:203-223) from the first rounds on
    "config2_k10_e0": (2, 100000, 10000, 0.0, 10, False),
    "config2_k10_e003": (2, 100000, 10000, 0.03, 10, False),
    "config2_k13_e0002": (2, 100000, 10000, 0.002, 13, False),
    "config2_k13_variable": (2, 100000, 10000, 0.0, 13, True),
    # SetIgnore (commands/overlap.go:203-223) only ever flags a read of at most two overlap sizes (2 000 bases) or one whose
    # length is within 10 % of the covered span - none of the 10 kb sets above has such reads.  1.2 Gbase of 1.5-4.5 kb reads
    # (served by the k-mer position index like config 2) flags thousands of them from the first round on
    "short_variable_k13": (2, 400000, 3000, 0.0, 13, True),
}


def map_config3():
    """BASELINE config 3: 50 000 reads x 8 kb (10 % error) mapped against a 4.6 Mb circular reference, k = 11 - the oracle's whole
    PAF (about two seconds of one core), kept as a hash for bench.py's map leg."""
    O.build_oracle()
    G, N, L, e, seed, k = 4600000, 50000, 8000, 0.1, 3, 11
    genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off = O.gen_reads(seed, G, N, L, e, False)
    t0 = time.time()
    paf, err = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=500, himem=False), circular=True, k=k)
    out = {"case": "config3_map", "generator": {"seed": seed, "genome": G, "reads": N, "read_len": L, "error": e, "variable": False},
           "k": k, "circular": True, "paf_lines": paf.count("\n"), "paf_sha256": hashlib.sha256(paf.encode()).hexdigest(),
           "stderr": err, "paf_head": paf.split("\n")[:3], "made_by": "tools/make_golden_full.py config3_map (oracle only)",
           "oracle_s": time.time() - t0}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden_full", "config3_map.json"), "w"), indent=1)
    print(json.dumps(out)[:600])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "config3_map":
        return map_config3()
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--rounds", type=int, default=-1)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--reads", type=int, default=0, help="override the read count (smaller surrogate, same generator)")
    a = ap.parse_args()
    seed, N, L, e, k, variable = CONFIGS[a.config]
    if a.reads:
        N = a.reads
    O.build_oracle()
    os.environ["DPO_SCAN_THREADS"] = str(a.threads)
    t0 = time.time()
    bases, off = O.gen_reads(seed, N * L // 20, N, L, e, variable)
    rs = O.ReadSet(bases, off, min_len=1000)
    t1 = time.time()
    run = O.OverlapRun(rs, k=k, max_rounds=a.rounds, traces=False)
    t2 = time.time()
    paf = run.paf
    out = {"case": a.config + ("" if a.rounds < 0 else "_first_%d_rounds" % a.rounds) + ("_%dreads" % N if a.reads else ""),
           "generator": {"seed": seed, "genome": N * L // 20, "reads": N, "read_len": L, "error": e, "variable": variable},
           "k": k, "rounds": run.rounds, "max_rounds": a.rounds, "paf_lines": paf.count("\n"),
           "paf_sha256": hashlib.sha256(paf.encode()).hexdigest(),
           "ignore_sha256": hashlib.sha256(rs.ignore().tobytes()).hexdigest(), "ignored_reads": int(rs.ignore().sum()),
           "paf_head": paf.split("\n")[:4],
           "made_by": "tools/make_golden_full.py (oracle only, own value table)", "oracle_s": t2 - t1, "generate_s": t1 - t0,
           "oracle_scan_threads": a.threads}
    path = os.path.join(ROOT, "tests", "golden_full", out["case"] + ".json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
