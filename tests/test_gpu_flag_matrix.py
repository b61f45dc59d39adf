"""The flag surface of the two commands on the device (missing 2): every flag of commands/overlap.go:23-26
(`overlap_size`, `num_seeds`, `chunk_size`, `query_batch_size`, `min_hits`) and commands/map.go:18-21 (`query_size`, `min_length`,
`chunk_size`, `seed_rate`) with values on both sides of its default, at k = 10 (dense seeds: every read is indexed and chopped)
and k = 13 (sparse), with one executor slot and with five.  The product must print the ORACLE's PAF round by round and flag the
same reads.  What the values move inside the kernels:
  num_seeds 30 / min_hits 0.4   minCount > 4: the 16-ladder (query_kernel<heavy>) and its step-8 omission inside `overlap`
  num_seeds 10                  what commands/correct.go:95-97 passes (minSeeds = 10)
  overlap_size 2000             twice the seeds per consensus sequence, other LDS layouts of consensus_full_kernel
  overlap_size 500              short windows, more queries per round
  chunk_size 5000 / 20000       chunk_kernel's piece length and overlap back-off (overlap.go:263-314)
  query_batch_size 50           the seed budget is never reached, a round ends by query count (overlap.go:57-60)
A value beyond a kernel's capacity must come back as DP_ERR_CAPACITY, never as a wrong PAF (the last test); query windows of more than
512 usable seeds - a capacity until round 5 - are a parity case since round 6."""
import ctypes as C

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

OVERLAP_FLAGS = [dict(overlap_size=500), dict(overlap_size=2000), dict(num_seeds=10), dict(num_seeds=30), dict(chunk_size=5000),
                 dict(chunk_size=20000), dict(min_hits=0.1), dict(min_hits=0.4), dict(query_batch_size=50),
                 dict(num_seeds=30, min_hits=0.4, overlap_size=2000), dict(num_seeds=10, chunk_size=5000, overlap_size=500),
                 # what `downpore correct` asks of the same Overlapper: QueryAll windows with minSeeds = 10 (commands/correct.go:95-97,169)
                 dict(num_seeds=10, query_type=4)]
INPUTS = {10: (110, 100000, 400, 5000, 0.01, True, 8), 13: (113, 1500000, 3000, 10000, 0.0, True, 5)}


def first_diff(a, b):
    if a == b:
        return None
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got  %s\n  want %s" % (i, x, y)
    return "line counts differ: got %d want %d" % (len(la), len(lb))


def _id(kw):
    return "-".join("%s=%s" % (a, b) for a, b in kw.items())


def _product_vs(orun, bases, off, rs, k, slots, max_rounds, kw):
    from downpore_amd.overlap import OverlapPipeline, Reads
    reads = Reads(bases, off, min_len=kw.get("overlap_size", 1000))
    pipe = OverlapPipeline(reads, k=k, slots=slots, **kw)
    rounds = 0
    limit = min(max_rounds, orun.rounds)
    pipe.H.dph_overlap_set_round_limit.restype = None
    pipe.H.dph_overlap_set_round_limit.argtypes = [C.c_void_p, C.c_int64]
    pipe.H.dph_overlap_set_round_limit(pipe.h, limit)  # (five slots commit every finished round they find: stop them at the oracle's last)
    while rounds < limit:
        c = pipe.step()
        if c == 0:
            break
        assert rounds + c <= orun.rounds, "the product ran more rounds than the oracle"
        want = "".join(orun.trace_paf(r) for r in range(rounds, rounds + c))
        d = first_diff(pipe.round_paf(), want)
        assert d is None, "slots=%d: PAF differs in rounds %d..%d: %s" % (slots, rounds, rounds + c - 1, d)
        rounds += c
    assert rounds >= limit, "slots=%d: the product stopped after %d of %d rounds" % (slots, rounds, limit)
    flagged = sorted(set(int(x) for r in range(rounds) for x in orun.trace(r, "newlyIgnored")))
    assert sorted(np.nonzero(reads.ignore())[0].tolist()) == flagged
    st = pipe.stats_total()
    pipe.close()
    return rounds, st


@pytest.mark.parametrize("k", [10, 13])
@pytest.mark.parametrize("kw", OVERLAP_FLAGS, ids=_id)
def test_overlap_flag(k, kw):
    seed, G, N, L, e, var, max_rounds = INPUTS[k]
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000))
    # a step may commit several rounds: the oracle runs a few rounds further than the product is asked to
    orun = O.OverlapRun(rs, k=k, max_rounds=max_rounds + 8, traces=True, **kw)
    assert orun.rounds >= 1
    lines = 0
    for slots in (1, 5):
        rounds, st = _product_vs(orun, bases, off, rs, k, slots, max_rounds, kw)
        lines = sum(orun.trace_paf(r).count("\n") for r in range(rounds))
    assert lines > 0, "a flag case without a single PAF line proves nothing"


@pytest.fixture(scope="module")
def ctx():
    import downpore_amd
    c = downpore_amd.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("k", [10, 13])
@pytest.mark.parametrize("kw", [dict(num_seeds=30, min_hits=0.4), dict(num_seeds=30, min_hits=0.4, overlap_size=2000),
                                dict(num_seeds=10, min_hits=0.1), dict(overlap_size=2000, chunk_size=5000),
                                dict(num_seeds=40, min_hits=0.45, overlap_size=3000)], ids=_id)
def test_index_query_chain_with_flags(ctx, k, kw):
    """The stages below the pipeline, through the kernel C ABI: the oracle's indexed chunks go into dp_index_build, its query
    windows into dp_find_overlaps with the flag's hit fraction; Matches() candidates (A14 / A5) and the chains (A6 - A8) must be
    the oracle's, entry for entry.  num_seeds 30 / 40 with min_hits 0.4 / 0.45 asks for 12 - 18 of 30 - 40 seeds: GetSharedIDs'
    16-ladder (util/bitset.go:308-411) - query_kernel<heavy> - inside `overlap`, the regime only `map` windows reached before."""
    seed, G, N, L, e, var, _ = INPUTS[k]
    bases, off = O.gen_reads(seed, G, N, L, e, var)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000))
    run = O.OverlapRun(rs, k=k, values=rs.kmer_values(k), max_rounds=2, traces=True, **kw)
    ctx.upload_reads(bases, off)
    matches = 0
    for rnd in range(run.rounds):
        ctx.round_begin(k, run.trace(rnd, "seedKmers"))
        qsegs, qoffs = run.trace(rnd, "querySegments")
        isegs, ioffs = run.trace(rnd, "indexedSegments")
        if len(ioffs) < 2:
            continue
        ctx.import_segments(isegs)
        ctx.index_build(ioffs[:-1].astype(np.uint64), ((ioffs[1:] - ioffs[:-1]) // 2).astype(np.uint32))
        out = ctx.find_overlaps(qsegs, qoffs.astype(np.uint64), kw.get("min_hits", 0.25), k, kw.get("overlap_size", 1000) // 2,
                                want_candidates=True)  # NewSeedAligner(lap.overlap/2), overlap/overlap.go:349
        cdata, coffs = run.trace(rnd, "candidates")
        assert np.array_equal(out["cand_off"].astype(np.int64), coffs), rnd
        assert np.array_equal(out["cand"].astype(np.int64), cdata), rnd
        ma, mao = run.trace(rnd, "matchA")
        mb, _ = run.trace(rnd, "matchB")
        for key, want in (("query", run.trace(rnd, "matchQueryIndex")), ("target", run.trace(rnd, "matchTarget")), ("off", mao),
                          ("match_a", ma), ("match_b", mb)):
            assert np.array_equal(np.asarray(out[key]).astype(np.int64), want), (rnd, key)
        matches += len(mao) - 1
    assert matches > 0


# ---------------------------------------------------------------------------------------------------------------- map
MAP_FLAGS = [dict(query_size=500), dict(query_size=2000), dict(seed_rate=20), dict(seed_rate=80), dict(chunk_size=5000),
             dict(chunk_size=20000), dict(min_length=2000), dict(query_size=500, seed_rate=20, chunk_size=5000, min_length=2000)]


@pytest.mark.parametrize("k", [11, 13])
@pytest.mark.parametrize("kw", MAP_FLAGS, ids=_id)
def test_map_flag(k, kw):
    """commands/map.go:18-21.  query_size moves the windows map_kernel scans and chains (mapping.go:489-589), seed_rate the
    AddSingleSeeds windows of the reference (seeds.go:140-200), chunk_size the reference chunks behind Matches(), min_length the
    reads that are mapped at all (commands/map.go:84-87)."""
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    G, N = 400000, 400
    genome = np.frombuffer(O.gen_genome(21, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off = O.gen_reads(21, G, N, 3600, 0.05, True)  # 1.8-5.4 kb: below min_length 2000, below 2 x query_size 2000
    ml = kw.get("min_length", 500)
    want, werr = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=ml, himem=False),
                           circular=True, k=k, **kw)
    got, gerr, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=ml, himem=False),
                              circular=True, k=k, **kw)
    d = first_diff(got, want)
    assert d is None, d
    assert gerr == werr
    assert want.count("\n") > N // 3


# ------------------------------------------------------------------------------------------- values beyond a kernel's capacity
def test_overlap_size_beyond_the_query_kernels_lds_lists():
    """query_kernel holds 512 posting sets per query in LDS (Q_MAXSETS, dp_overlap.hip); the reference has no such limit (allSeedSets
    grows, seeds/seeds.go:336-347).  A 6 000-base window over a genome the round's seeds cover three times has ~900 usable seeds: until
    round 5 that came back as DP_ERR_CAPACITY, since round 6 the BIG variants of the kernel keep such a query's lists in global memory -
    `-overlap_size 6000 -num_seeds 60` must print the oracle's PAF, round by round, with one slot and with five."""
    seed, G, N, L, e, var, _ = INPUTS[10]
    bases, off = O.gen_reads(seed, G, N, 14000, e, False)
    kw = dict(overlap_size=6000, num_seeds=60)
    rs = O.ReadSet(bases, off, min_len=6000)
    orun = O.OverlapRun(rs, k=10, max_rounds=4 + 8, traces=True, **kw)
    assert orun.rounds >= 1
    n_seeds_in_queries = max(len(orun.trace(r, "querySegments")) for r in range(min(orun.rounds, 4)))
    assert n_seeds_in_queries > 0
    lines = 0
    for slots in (1, 5):
        rounds, st = _product_vs(orun, bases, off, rs, 10, slots, 4, kw)
        lines = sum(orun.trace_paf(r).count("\n") for r in range(rounds))
    assert lines > 0


def test_map_query_size_beyond_the_map_kernels_lds_arrays():
    """`map -query_size 8000 -seed_rate 10`: ~800 seeds per window - more than query_kernel's LDS lists (512 sets) and map_kernel's LDS
    arrays (256 reduced window seeds) hold.  DP_ERR_CAPACITY until round 5; since round 6 both kernels have a variant with the working
    set in global memory (query_kernel<.., BIG>, map_kernel<BIG>): the command must print the oracle's PAF and stderr."""
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    G, N = 300000, 50
    genome = np.frombuffer(O.gen_genome(23, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off = O.gen_reads(23, G, N, 20000, 0.02, False)
    kw = dict(circular=True, k=11, query_size=8000, seed_rate=10)
    want, werr = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=500, himem=False), **kw)
    got, gerr, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=500, himem=False), **kw)
    assert first_diff(got, want) is None
    assert gerr == werr
    assert want.count("\n") >= N // 2
