"""Isolates a flag-matrix failure: runs one flag set through the product under a few switches (each in a fresh process) and
through the staged kernel comparison (index -> candidates -> chains against the oracle's trace).
usage: python tools/r05/debug_flags.py k '{"num_seeds":30,...}' [mode]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as O  # noqa: E402

INPUTS = {10: (110, 100000, 400, 5000, 0.01, True, 8), 13: (113, 1500000, 3000, 10000, 0.0, True, 5)}


def first_diff(a, b):
    if a == b:
        return None
    la, lb = a.split("\n"), b.split("\n")
    nd = sum(1 for x, y in zip(la, lb) if x != y)
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d of %d/%d (%d differ):\n  got  %s\n  want %s" % (i, len(la), len(lb), nd, x, y)
    return "line counts differ: got %d want %d" % (len(la), len(lb))


def e2e(k, kw, rounds=2):
    from downpore_amd.overlap import OverlapPipeline, Reads
    seed, G, N, L, e, var, _ = INPUTS[k]
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000))
    orun = O.OverlapRun(rs, k=k, max_rounds=rounds, traces=True, **kw)
    reads = Reads(bases, off, min_len=kw.get("overlap_size", 1000))
    pipe = OverlapPipeline(reads, k=k, slots=1, **kw)
    pipe.H.dph_overlap_set_round_limit.restype = None
    import ctypes as C
    pipe.H.dph_overlap_set_round_limit.argtypes = [C.c_void_p, C.c_int64]
    pipe.H.dph_overlap_set_round_limit(pipe.h, rounds)
    n = 0
    while n < rounds:
        c = pipe.step()
        if c == 0:
            break
        st = pipe.stats()
        got = pipe.round_paf()
        want = "".join(orun.trace_paf(r) for r in range(n, n + c))
        print("   rounds %d..%d: product queries %d indexed %d matches %d paf %d | oracle queries %d indexed %d matches %d paf %d" % (
            n, n + c - 1, st["n_queries"], st["n_indexed"], st["n_matches"], got.count("\n"),
            len(orun.trace(n + c - 1, "queryIDs")), len(orun.trace(n + c - 1, "indexedIds")), len(orun.trace(n + c - 1, "matchTarget")), want.count("\n")))
        gs, ws = set(got.split("\n")), set(want.split("\n"))
        only_w = [x for x in want.split("\n") if x not in gs]
        only_g = [x for x in got.split("\n") if x not in ws]
        print("     lines only in oracle %d, only in product %d" % (len(only_w), len(only_g)))
        for x in only_w[:6]:
            print("       want:", x)
        for x in only_g[:6]:
            print("       got: ", x)
        n += c
    pipe.close()


def staged(k, kw, rounds=2):
    import downpore_amd
    seed, G, N, L, e, var, _ = INPUTS[k]
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000))
    values = rs.kmer_values(k)
    run = O.OverlapRun(rs, k=k, values=values, max_rounds=rounds, traces=True, **kw)
    ctx = downpore_amd.Context(0)
    ctx.upload_reads(bases, off)
    for rnd in range(run.rounds):
        seed_kmers = run.trace(rnd, "seedKmers")
        ctx.round_begin(k, seed_kmers)
        qsegs, qoffs = run.trace(rnd, "querySegments")
        isegs, ioffs = run.trace(rnd, "indexedSegments")
        ctx.import_segments(isegs)
        nseeds = ((ioffs[1:] - ioffs[:-1]) // 2).astype(np.uint32)
        ctx.index_build(ioffs[:-1].astype(np.uint64), nseeds)
        out = ctx.find_overlaps(qsegs, qoffs.astype(np.uint64), kw.get("min_hits", 0.25), k, kw.get("overlap_size", 1000) // 2, want_candidates=True)
        cdata, coffs = run.trace(rnd, "candidates")
        print("   round %d: %d seeds, %d indexed, %d queries (max %d seeds), %d candidates, %d matches" % (
            rnd, len(seed_kmers), len(ioffs) - 1, len(qoffs) - 1, int(((qoffs[1:] - qoffs[:-1]) // 2).max()), len(cdata),
            len(run.trace(rnd, "matchTarget"))))
        ok = np.array_equal(out["cand_off"].astype(np.int64), coffs) and np.array_equal(out["cand"].astype(np.int64), cdata)
        print("     candidates:", "EQUAL" if ok else "DIFFER (got %d want %d)" % (len(out["cand"]), len(cdata)))
        if not ok:
            for q in range(len(coffs) - 1):
                a = out["cand"][int(out["cand_off"][q]):int(out["cand_off"][q + 1])].astype(np.int64)
                b = cdata[coffs[q]:coffs[q + 1]]
                if not np.array_equal(a, b):
                    print("     first differing query %d (%d seeds): got %s want %s" % (q, (qoffs[q + 1] - qoffs[q]) // 2, a[:12], b[:12]))
                    break
        mq, mt = run.trace(rnd, "matchQueryIndex"), run.trace(rnd, "matchTarget")
        ma, mao = run.trace(rnd, "matchA")
        mb, _ = run.trace(rnd, "matchB")
        for key, want in (("query", mq), ("target", mt), ("off", mao), ("match_a", ma), ("match_b", mb)):
            got = np.asarray(out[key]).astype(np.int64)
            if not np.array_equal(got, want):
                n = min(len(got), len(want))
                bad = np.nonzero(got[:n] != want[:n])[0]
                print("     %s DIFFER: got %d want %d entries, first at %s" % (key, len(got), len(want), bad[:1]))
            else:
                print("     %s EQUAL" % key)


if __name__ == "__main__":
    k = int(sys.argv[1])
    kw = json.loads(sys.argv[2])
    mode = sys.argv[3] if len(sys.argv) > 3 else "all"
    if mode == "e2e":
        e2e(k, kw)
    elif mode == "staged":
        staged(k, kw)
    else:
        for env in ({}, {"DP_DEVICE_CONSENSUS": "0", "DP_DEVICE_CHUNK": "0", "DP_FIND_PENDING": "0", "DP_QUERY_PRESTAGE": "0"}):
            print("== e2e with", env, flush=True)
            subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], "e2e"], env=dict(os.environ, **env))
        print("== staged", flush=True)
        subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], "staged"])
