"""CPU test of the PRODUCT's host consensus/PAF stage (libdownpore_host.so, no GPU involved): fed with the ORACLE's
per-round queries / indexed sequences / matches it must print the oracle's PAF and flag the same reads."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O


def first_diff(a, b):
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got  %s\n  want %s" % (i, x, y)
    if len(la) != len(lb):
        return "line counts differ: got %d want %d" % (len(la), len(lb))
    return None


def run_finalcheck(H, reads, k, overlap_size, run, rnd):
    i64, i32 = np.int64, np.int32
    seeds = run.trace(rnd, "seedKmers").astype(np.uint32)
    qd, qo = run.trace(rnd, "querySegments")
    idd, io = run.trace(rnd, "indexedSegments")
    ma, mo = run.trace(rnd, "matchA")
    mb, _ = run.trace(rnd, "matchB")
    arrs = dict(qsegs=qd.astype(i32), qoff=qo.astype(i64), qid=run.trace(rnd, "queryIDs"), qseq=run.trace(rnd, "querySeqIDs"),
                qlen=run.trace(rnd, "queryLength"), qoffset=run.trace(rnd, "queryOffset"), qinset=run.trace(rnd, "queryInset"),
                isegs=idd.astype(i32), ioff=io.astype(i64), iid=run.trace(rnd, "indexedIds"), ilen=run.trace(rnd, "indexedLength"),
                ioffset=run.trace(rnd, "indexedOffset"), iinset=run.trace(rnd, "indexedInset"),
                mq=run.trace(rnd, "matchQueryIndex"), mt=run.trace(rnd, "matchTarget"), moff=mo.astype(i64), ma=ma.astype(i32),
                mb=mb.astype(i32))
    arrs = {k_: np.ascontiguousarray(v) for k_, v in arrs.items()}
    sc = run.trace(rnd, "scalars")
    n = C.c_int64(0)
    stats = np.zeros(3, dtype=i64)
    H.dph_finalcheck.restype = C.POINTER(C.c_char)
    H.dph_finalcheck.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64] + [C.c_void_p] * 7 + [C.c_int64] + \
        [C.c_void_p] * 6 + [C.c_int64] + [C.c_void_p] * 5 + [C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]
    p = H.dph_finalcheck(reads.h, k, overlap_size, seeds.ctypes.data, len(seeds), arrs["qsegs"].ctypes.data,
                         arrs["qoff"].ctypes.data, arrs["qid"].ctypes.data, arrs["qseq"].ctypes.data, arrs["qlen"].ctypes.data,
                         arrs["qoffset"].ctypes.data, arrs["qinset"].ctypes.data, len(arrs["qid"]), arrs["isegs"].ctypes.data,
                         arrs["ioff"].ctypes.data, arrs["iid"].ctypes.data, arrs["ilen"].ctypes.data, arrs["ioffset"].ctypes.data,
                         arrs["iinset"].ctypes.data, len(arrs["iid"]), arrs["mq"].ctypes.data, arrs["mt"].ctypes.data,
                         arrs["moff"].ctypes.data, arrs["ma"].ctypes.data, arrs["mb"].ctypes.data, len(arrs["mq"]), int(sc[1]),
                         C.byref(n), stats.ctypes.data)
    return C.string_at(p, n.value).decode(), stats


@pytest.mark.parametrize("seed,G,N,L,k,e,variable", [(110, 100000, 400, 5000, 10, 0.0, False),
                                                      (110, 80000, 300, 6000, 10, 0.03, True),
                                                      (1, 250000, 1000, 5000, 10, 0.0, False)])
def test_product_finalcheck_matches_oracle(seed, G, N, L, k, e, variable):
    from downpore_amd.overlap import Reads, load_host
    H = load_host()
    bases, off = O.gen_reads(seed, G, N, L, e, variable)
    rs = O.ReadSet(bases, off, min_len=1000)
    run = O.OverlapRun(rs, k=k, max_rounds=3, traces=True)
    reads = Reads(bases, off, min_len=1000)
    for rnd in range(run.rounds):
        paf, stats = run_finalcheck(H, reads, k, 1000, run, rnd)
        want = run.trace_paf(rnd)
        d = first_diff(paf, want)
        assert d is None, "round %d %s" % (rnd, d)
    assert np.array_equal(reads.ignore(), rs.ignore())


@pytest.mark.parametrize("k,G,N,L", [(8, 30000, 200, 3000), (10, 200000, 500, 4000), (11, 400000, 300, 6000)])
def test_product_value_table_matches_oracle(k, G, N, L):
    """kmerValuesFromCounts (histogram -> value table, incl. the top-1 % blacklist and its tie rule) against the oracle,
    from the oracle's own k-mer counts."""
    from downpore_amd.overlap import load_host
    H = load_host()
    bases, off = O.gen_reads(40 + k, G, N, L, 0.01, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    counts = rs.kmer_counts(k)
    want = rs.kmer_values(k)
    got = np.zeros(4 ** k, dtype=np.float64)
    c = np.ascontiguousarray(counts, dtype=np.uint64).copy()
    H.dph_values_from_counts.restype = None
    H.dph_values_from_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    H.dph_values_from_counts(c.ctypes.data, k, got.ctypes.data)
    assert np.array_equal(got, want)
    assert (got > 0).sum() > 0
