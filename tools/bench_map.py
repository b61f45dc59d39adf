#!/usr/bin/env python3
"""Times `downpore map` on the GPU for BASELINE config 3 (E. coli scale: 50k reads x 8 kb against a 4.6 Mb synthetic
circular reference, k=11) and, on a bounded sample, the oracle (CPU port) next to it.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=4600000)
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--read-len", type=int, default=8000)
    ap.add_argument("--error", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--cpu-reads", type=int, default=1500)
    ap.add_argument("--k", type=int, default=11)
    a = ap.parse_args()
    from tools.synth import gen_genome, gen_reads
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    genome = np.frombuffer(gen_genome(a.seed, a.genome), dtype=np.uint8)
    goff = np.array([0, a.genome], dtype=np.int64)
    bases, off = gen_reads(a.seed, a.genome, a.reads, a.read_len, a.error, False)
    ref = Reads(genome, goff, min_len=0, himem=False)
    reads = Reads(bases, off, min_len=500, himem=False)
    t0 = time.perf_counter()
    paf, err, st = map_reads(ref, reads, circular=True, k=a.k)
    dt = time.perf_counter() - t0
    lines = paf.count("\n")
    out = {"workload": "map: %d reads x %d bp (error %.2f) vs %d bp circular reference, k=%d" %
                       (a.reads, a.read_len, a.error, a.genome, a.k),
           "wall_s": dt, "reads_per_s": a.reads / dt, "paf_lines": lines, "stats": st,
           "stderr": err.strip().split("\n")[-4:]}
    if a.cpu_reads > 0:
        from tests import oracle_lib as O
        n = min(a.cpu_reads, a.reads)
        oref = O.ReadSet(genome, goff, min_len=0, himem=False)
        oreads = O.ReadSet(bases[:off[n]], off[:n + 1], min_len=500, himem=False)
        t0 = time.perf_counter()
        opaf, oerr = O.map_run(oref, oreads, circular=True, k=a.k)
        odt = time.perf_counter() - t0
        out["cpu_baseline"] = {"reads_per_s": n / odt, "sample": "first %d reads (incl. reference indexing), %.1f s" % (n, odt),
                               "cores": 1, "kind": "port", "identical_prefix": paf.startswith(opaf)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
