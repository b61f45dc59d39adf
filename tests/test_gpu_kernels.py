"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP stage called through the C ABI
(include/downpore_hip.h) is compared bit-for-bit with the ORACLE's per-round trace on the same seeded inputs.
"""
import os

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import downpore_amd
    c = downpore_amd.Context(0)
    yield c
    c.close()


def _case(seed, G, N, L, k, e=0.0, variable=False, rounds=2, **kw):
    bases, off = O.gen_reads(seed, G, N, L, e, variable)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000))
    values = rs.kmer_values(k)
    run = O.OverlapRun(rs, k=k, values=values, max_rounds=rounds, traces=True, **kw)
    return bases, off, rs, values, run


def _expected_segments(read_str, k, seed_kmers, start=None, end=None):
    table = np.zeros(4 ** k, dtype=np.uint8)
    table[seed_kmers] = 1
    kmap = {int(km): i for i, km in enumerate(seed_kmers)}
    s = O.Seq(read_str)
    v = s.sub(0 if start is None else start, len(read_str) if end is None else end)
    seg = v.write_segments(k, table)
    seg[1::2] = [kmap[int(x)] for x in seg[1::2]]
    return seg


def test_pack_matches_reference_encoding(ctx):
    bases, off = O.gen_reads(11, 20000, 37, 777, 0.0, True)
    ctx.upload_reads(bases, off)
    for r in (0, 1, 17, 36):
        s = bases[off[r]:off[r + 1]].tobytes().decode()
        want = O.Seq(s).bytes()
        got = ctx.packed_read(r)
        assert np.array_equal(want, got), r


def test_pack_reverse_complement_pairs(ctx):
    """dp_reads_upload_rc: the device-made reverse strands equal the oracle's packed ReverseComplement()."""
    reads = ["ACGTTGCAAGGCTTAACCGGA", "T" * 33 + "ACG", "GATTACA" * 9 + "C", "ACGT" * 16, "NNACGTRYACG"]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.cumsum([0] + [len(r) for r in reads]).astype(np.int64)
    ctx.upload_reads_rc(bases, off, 2)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    dev = 0
    for i, r in enumerate(reads):
        strands = [r] if i < 2 else [r, None]
        for st in strands:
            got = ctx.packed_read(dev)
            if st is None:  # 3 - code per base, reversed (what the host built before): compare via the oracle's packer
                codes = [3 - (((ord(c) >> 1) ^ ((ord(c) & 4) >> 2)) & 3) for c in reversed(r)]
                st = "".join("ACGT"[c] for c in codes)
            want = O.Seq(st).bytes()
            assert bytes(got) == bytes(want[:len(got)]), (i, dev)
            dev += 1


def test_packed_upload_places_reads_and_makes_reverse_strands(ctx):
    """dp_reads_upload_packed_rc (round 6): the host hands the reads over 2-bit packed, every read on a 16-byte boundary; the device puts
    them in place and makes the reverse strands from the packed forward ones.  Every device read must equal what dp_reads_upload_rc
    makes from the ASCII - lengths around every multiple of 4, 16 and 64, empty reads, any byte as a base, garbage in the padding and in
    the last byte's unused bits, from pageable memory (staged) and from a block of dp_host_alloc (one copy)."""
    rng = np.random.default_rng(5)
    lens = [0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 66, 67, 68, 127, 128, 129, 255, 256, 257, 1000, 4099, 0, 7] + list(rng.integers(1, 3000, 60))
    alphabet = np.frombuffer(b"ACGTacgtNRY", dtype=np.uint8)
    reads = [alphabet[rng.integers(0, len(alphabet), n)] if i % 3 else rng.integers(0, 256, n, dtype=np.uint8) for i, n in enumerate(lens)]
    bases = np.concatenate(reads).astype(np.uint8)
    off = np.cumsum([0] + lens).astype(np.int64)
    for first_paired in (2, 0, len(lens)):
        ctx.upload_reads_rc(bases, off, first_paired)
        n_dev = first_paired + 2 * (len(lens) - first_paired)
        want = [bytes(ctx.packed_read(d)) for d in range(n_dev)]
        poff = [0]
        for n in lens:
            poff.append(poff[-1] + (((n + 3) // 4 + 15) & ~15))
        packed = rng.integers(0, 256, poff[-1], dtype=np.uint8)   # garbage wherever no base is
        for r, b in enumerate(reads):
            n = len(b)
            if n:
                pk = np.array(O.Seq(h=O.lib().dpo_seq_new(b.tobytes(), n)).bytes())
                if n % 4:
                    pk[-1] |= int(rng.integers(0, 256)) & ((1 << (2 * (4 - n % 4))) - 1)
                packed[poff[r]:poff[r] + len(pk)] = pk
        for pinned in (False, True):
            ctx.upload_reads_packed_rc(packed, np.array(lens, dtype=np.uint32), first_paired, pinned=pinned)
            got = [bytes(ctx.packed_read(d)) for d in range(n_dev)]
            bad = [d for d in range(n_dev) if got[d] != want[d]]
            assert not bad, (first_paired, pinned, bad[:5])


def test_histogram(ctx):
    bases, off = O.gen_reads(12, 50000, 200, 1500, 0.01, True)
    ctx.upload_reads(bases, off)
    rs = O.ReadSet(bases, off, min_len=0)
    for k in (6, 10):
        assert np.array_equal(ctx.kmer_histogram(k), rs.kmer_counts(k))


@pytest.mark.parametrize("k,G,N,L,e", [(3, 20000, 100, 1500, 0.0), (6, 50000, 200, 1500, 0.01), (8, 30000, 200, 3000, 0.01),
                                       (10, 200000, 500, 4000, 0.01), (11, 400000, 300, 6000, 0.0),
                                       (12, 1000000, 2000, 10000, 0.0), (13, 1500000, 3000, 10000, 0.0)])
def test_value_table_on_device(ctx, k, G, N, L, e):
    """dp_kmer_values (histogram, value per k-mer, fwd+rc merge, top-1 % cut with its tie rule, all on the device) is
    bit-identical to the oracle's table (overlap.go:55-93, kmers.go:87-112) - even k has palindromic k-mers, every
    table is full of tied counts, k=3 is below the n/100 cut - and it is the table dp_select_seeds then uses."""
    bases, off = O.gen_reads(60 + k, G, N, L, e, True)
    ctx.upload_reads(bases, off)
    rs = O.ReadSet(bases, off, min_len=0)
    want = rs.kmer_values(k)
    got = ctx.kmer_values(k)
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert (got > 0).sum() > 0
    if k >= 8:  # the resident copy: selection with it equals selection with an uploaded oracle table
        wins = [(r, 0, min(int(off[r + 1] - off[r]), 1000)) for r in range(0, N, max(1, N // 40))]
        a = ctx.select_seeds(wins, k, 15)
        ctx.values_upload(want)
        b = ctx.select_seeds(wins, k, 15)
        assert np.array_equal(a, b)


def test_value_table_no_reads(ctx):
    ctx.upload_reads(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
    assert not ctx.kmer_values(6).any()


@pytest.mark.parametrize("k,G,N,L,e", [(10, 100000, 400, 5000, 0.0), (10, 60000, 300, 4000, 0.02),
                                       (13, 1500000, 3000, 10000, 0.0)])
def test_scan_index_query_chain(ctx, k, G, N, L, e):
    bases, off, rs, values, run = _case(21 + k, G, N, L, k, e, rounds=2)
    assert run.rounds >= 1
    ctx.upload_reads(bases, off)
    ignore = np.zeros(N, dtype=np.uint8)
    for rnd in range(run.rounds):
        seed_kmers = run.trace(rnd, "seedKmers")
        ctx.round_begin(k, seed_kmers)

        # ---- A2 + A10: whole reads (cached views) and the query windows
        reads = [r for r in range(N) if not ignore[r]]
        sample = reads[:: max(1, len(reads) // 60)]
        items = [(r, 0, int(off[r + 1] - off[r]) - k + 1, 0) for r in sample]
        qsegs, qoffs = run.trace(rnd, "querySegments")
        qseq = run.trace(rnd, "querySeqIDs")
        # forward queries are the even entries; windows = first/last 1000 bases of each read >= 2000 (overlap.go:61-80)
        qitems, qexp = [], []
        seen = {}
        for qi in range(0, len(qseq), 2):
            r = int(qseq[qi])
            Lr = int(off[r + 1] - off[r])
            n = seen.get(r, 0)
            seen[r] = n + 1
            if Lr < 2000:
                st, en = 0, Lr
            else:
                st, en = (0, 1000) if n == 0 else (Lr - 1000, Lr)
            qitems.append((r, st, en - st - k + 1, 0))
            qexp.append(qsegs[qoffs[qi]:qoffs[qi + 1]])
        res = ctx.scan(items + qitems)
        so, sg = res["seg_off"], res["segs"]
        for n, r in enumerate(sample):
            want = _expected_segments(bases[off[r]:off[r + 1]].tobytes().decode(), k, seed_kmers)
            got = sg[int(so[n]):int(so[n + 1])]
            assert np.array_equal(want, got), ("read", r)
            assert res["n_seeds"][n] == len(want) // 2
        base = len(sample)
        for n, want in enumerate(qexp):
            got = sg[int(so[base + n]):int(so[base + n + 1])]
            assert np.array_equal(want, got), ("query", n)

        # min_seeds filter: counts reported, segments withheld
        res2 = ctx.scan([(r, 0, int(off[r + 1] - off[r]) - k + 1, 15) for r in sample])
        for n, r in enumerate(sample):
            c = int(res2["n_seeds"][n])
            ln = int(res2["seg_off"][n + 1] - res2["seg_off"][n])
            assert ln == (2 * c + 1 if c >= 15 else 0)

        # ---- A13: index over the oracle's indexed (chunked) sequences
        isegs, ioffs = run.trace(rnd, "indexedSegments")
        M = len(ioffs) - 1
        ctx.import_segments(isegs)
        nseeds = ((ioffs[1:] - ioffs[:-1]) // 2).astype(np.uint32)
        ctx.index_build(ioffs[:-1].astype(np.uint64), nseeds)
        S = len(seed_kmers)
        post = [set() for _ in range(S)]
        for i in range(M):
            for s in isegs[ioffs[i] + 1:ioffs[i + 1]:2]:
                post[int(s)].add(i)
        for s in list(range(0, S, max(1, S // 40))):
            words, cnt, st, en = ctx.posting_row(s)
            ids = sorted(post[s])
            got = [w * 64 + b for w in range(len(words)) for b in range(64) if (int(words[w]) >> b) & 1]
            assert got == ids
            assert cnt == len(ids)
            if ids:
                assert (st, en) == (ids[0] // 64, ids[-1] // 64)
            else:
                assert (st, en) == (1, 0)
        for i in range(0, M, max(1, M // 20)):
            words = ctx.seedset_row(i)
            got = sorted(w * 64 + b for w in range(len(words)) for b in range(64) if (int(words[w]) >> b) & 1)
            assert got == sorted(set(int(x) for x in isegs[ioffs[i] + 1:ioffs[i + 1]:2]))

        # ---- A14/A5 candidates, A6/A7/A8 matches
        out = ctx.find_overlaps(qsegs, qoffs.astype(np.uint64), 0.25, k, 500, want_candidates=True)
        cdata, coffs = run.trace(rnd, "candidates")
        assert np.array_equal(out["cand_off"].astype(np.int64), coffs)
        assert np.array_equal(out["cand"].astype(np.int64), cdata)
        mq = run.trace(rnd, "matchQueryIndex")
        mt = run.trace(rnd, "matchTarget")
        ma, mao = run.trace(rnd, "matchA")
        mb, _ = run.trace(rnd, "matchB")
        assert np.array_equal(out["query"].astype(np.int64), mq)
        assert np.array_equal(out["target"].astype(np.int64), mt)
        assert np.array_equal(out["off"].astype(np.int64), mao)
        assert np.array_equal(out["match_a"].astype(np.int64), ma)
        assert np.array_equal(out["match_b"].astype(np.int64), mb)
        # target anchors of every chain: GetSeedOffset(first matched target seed), GetSeedOffsetFromEnd(last one)
        anchors = out["target_anchor"]
        assert anchors.shape == (len(mt), 2)
        for i in range(len(mt)):
            seg = isegs[ioffs[int(mt[i])]:ioffs[int(mt[i]) + 1]].astype(np.int64)
            first, last = int(mb[mao[i]]), int(mb[mao[i + 1] - 1])
            want_first = int(seg[0] + (seg[2:2 * first + 1:2] + k).sum())
            want_last = int(seg[-1] + (seg[2 * last + 2:len(seg) - 1:2] + k).sum())
            assert (int(anchors[i, 0]), int(anchors[i, 1])) == (want_first, want_last), ("anchor", i)
        # the chain kernel's slower tiers (open chains in LDS; one-lane transcription) must give the same chains
        for tier in ("2", "3"):
            os.environ["DP_CHAIN_TIER"] = tier
            try:
                o2 = ctx.find_overlaps(qsegs, qoffs.astype(np.uint64), 0.25, k, 500, want_candidates=False)
            finally:
                del os.environ["DP_CHAIN_TIER"]
            for key in ("query", "target", "off", "match_a", "match_b"):
                assert np.array_equal(np.asarray(o2[key]), np.asarray(out[key])), (tier, key)

        # the chaining stage with 0 (serial walk alone), 1 and 3 proposal passes instead of the default 2: same records
        for passes in ("0", "1", "3"):
            os.environ["DP_CHAIN_PASSES"] = passes
            try:
                o3 = ctx.find_overlaps(qsegs, qoffs.astype(np.uint64), 0.25, k, 500, want_candidates=False)
            finally:
                del os.environ["DP_CHAIN_PASSES"]
            for key in ("query", "target", "off", "match_a", "match_b", "target_anchor"):
                assert np.array_equal(np.asarray(o3[key]), np.asarray(out[key])), ("passes", passes, key)

        for r in run.trace(rnd, "newlyIgnored"):
            ignore[int(r)] = 1


def test_scan_top_level_len_mod4_quirk(ctx):
    """himem=false re-reads are top-level sequences: with len%4==0 the reference examines 4 fewer k-mers
    (SURVEY §8(a) A2(i)); the caller encodes that in n_kmers and the final gap follows."""
    k = 10
    bases, off = O.gen_reads(5, 30000, 50, 1200, 0.0, False)  # 1200 % 4 == 0
    ctx.upload_reads(bases, off)
    rng = np.random.default_rng(1)
    seed_kmers = np.unique(rng.integers(1, 4 ** k, 4000)).astype(np.int64)
    ctx.round_begin(k, seed_kmers)
    table = np.zeros(4 ** k, dtype=np.uint8)
    table[seed_kmers] = 1
    kmap = {int(km): i for i, km in enumerate(seed_kmers)}
    items = [(r, 0, 1200 - k + 1 - 4, 0) for r in range(50)]
    res = ctx.scan(items)
    for r in range(50):
        s = O.Seq(bases[off[r]:off[r + 1]].tobytes().decode())  # top level, finalLen == 0
        want = s.write_segments(k, table)
        want[1::2] = [kmap[int(x)] for x in want[1::2]]
        got = res["segs"][int(res["seg_off"][r]):int(res["seg_off"][r + 1])]
        assert np.array_equal(want, got)


def test_empty_and_ragged(ctx):
    k = 10
    bases, off = O.gen_reads(6, 30000, 20, 900, 0.0, True)
    ctx.upload_reads(bases, off)
    ctx.round_begin(k, np.zeros(0, dtype=np.uint32))
    res = ctx.scan(np.zeros((0, 4), dtype=np.uint32))
    assert len(res["n_seeds"]) == 0
    items = [(r, 0, int(off[r + 1] - off[r]) - k + 1, 0) for r in range(20)]
    res = ctx.scan(items)
    assert res["n_seeds"].sum() == 0
    for r in range(20):  # no seeds: single gap = bases scanned
        assert res["segs"][int(res["seg_off"][r])] == off[r + 1] - off[r]
    ctx.index_build(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    out = ctx.find_overlaps(np.array([5], dtype=np.int32), np.array([0, 1], dtype=np.uint64), 0.25, k, 500, True)
    assert len(out["query"]) == 0


def test_scan_prepare_index_and_scan_modes_agree(ctx):
    """dp_scan_prepare builds the k-mer position index ahead of the rounds (right after dp_kmer_values it starts from that
    call's histogram); dp_scan_reads then answers from it and returns byte for byte what the scan kernels return - also
    after the stream was moved to the highest priority (dp_ctx_set_priority) and across a mid-run switch of the mode."""
    k = 10
    bases, off = O.gen_reads(16, 80000, 300, 3000, 0.01, True)
    N = 300
    ctx.upload_reads(bases, off)
    ctx.kmer_values(k)  # leaves the histogram for the index build
    rng = np.random.default_rng(3)
    ignore = (rng.random(N) < 0.1).astype(np.uint8)
    extra = [(5, 100, 500, 0), (9, 0, 300, 0)]
    os.environ["DP_SCAN_INDEX"] = "1"
    try:
        ctx.scan_prepare(k)
        ctx.set_priority(True)
        outs = {}
        for rnd in range(3):
            seeds = np.unique(rng.integers(1, 4 ** k, 6000)).astype(np.uint32)
            ctx.round_begin(k, seeds)
            for mode in ("1", "0", "1"):
                os.environ["DP_SCAN_INDEX"] = mode
                got = ctx.scan_reads(ignore, 11, 0, N, False, 20, extra)
                assert got["index_mode"] == int(mode)
                key = (rnd, mode)
                if key in outs:  # the second pass in index mode after a scan-mode pass
                    prev = outs[key]
                    assert all(np.array_equal(prev[f], got[f]) for f in ("read", "n_seeds", "seg_off", "extra_n_seeds", "segs"))
                outs[key] = {f: np.array(got[f]).copy() for f in ("read", "n_seeds", "seg_off", "extra_n_seeds", "segs")}
            a, b = outs[(rnd, "1")], outs[(rnd, "0")]
            assert len(a["read"]) > 0
            for f in a:
                assert np.array_equal(a[f], b[f]), (rnd, f)
    finally:
        del os.environ["DP_SCAN_INDEX"]
        ctx.set_priority(False)


@pytest.mark.parametrize("n_seeds", [40000, 150000, 400000, 800000])
def test_index_mode_write_sort_tiers(ctx, n_seeds):
    """The index-mode write step sorts a read's hits inside one wave's LDS: shuffle ranks up to 64 hits, an LDS rank sort up to
    half the block's key capacity, a bitonic network above that (three block sizes, picked by the round's largest count).
    Denser and denser seed sets walk a 3000-base read through every tier; the scan kernels are the check."""
    k = 10
    bases, off = O.gen_reads(21, 60000, 200, 3000, 0.01, True)
    N = 200
    ctx.upload_reads(bases, off)
    rng = np.random.default_rng(n_seeds)
    seeds = rng.choice(np.arange(1, 4 ** k, dtype=np.uint32), size=n_seeds, replace=False)
    ignore = np.zeros(N, dtype=np.uint8)
    os.environ["DP_SCAN_INDEX"] = "1"
    try:
        ctx.scan_prepare(k)
        ctx.round_begin(k, np.sort(seeds))
        outs = {}
        for mode in ("1", "0"):
            os.environ["DP_SCAN_INDEX"] = mode
            got = ctx.scan_reads(ignore, 3, 0, N, False, 20, [(7, 10, 900, 0)])
            assert got["index_mode"] == int(mode)
            outs[mode] = {f: np.array(got[f]).copy() for f in ("read", "n_seeds", "seg_off", "extra_n_seeds", "segs")}
        assert len(outs["1"]["read"]) > 0
        for f in outs["1"]:
            assert np.array_equal(outs["1"][f], outs["0"][f]), f
        print("largest hit count", int(outs["1"]["n_seeds"].max()))
    finally:
        del os.environ["DP_SCAN_INDEX"]


def test_scan_reads_compaction_matches_itemwise_scan(ctx):
    """dp_scan_reads (items generated and survivors compacted on the device) == dp_scan item by item + host filter."""
    k = 10
    bases, off = O.gen_reads(15, 80000, 300, 3000, 0.01, True)
    N = 300
    ctx.upload_reads(bases, off)
    rng = np.random.default_rng(2)
    seeds = np.unique(rng.integers(1, 4 ** k, 9000)).astype(np.uint32)
    ctx.round_begin(k, seeds)
    ignore = (rng.random(N) < 0.2).astype(np.uint8)
    lens = np.diff(off)
    for top_level in (False, True):
        for (lo, hi) in ((0, N), (37, 211)):
            items = []
            for r in range(lo, hi):
                n = int(lens[r]) - k + 1
                if top_level and lens[r] % 4 == 0:
                    n -= 4
                items.append((r, 0, max(n, 0), 25))
            full = ctx.scan(items)
            extra = [(5, 100, 500, 0), (9, 0, 300, 0)]
            got = ctx.scan_reads(ignore, 7 if top_level else 3, lo, hi, top_level, 25, extra)
            want_reads = [r for i, r in enumerate(range(lo, hi)) if not ignore[r] and full["n_seeds"][i] >= 25]
            assert list(got["read"]) == want_reads
            for j, r in enumerate(want_reads):
                i = r - lo
                a = full["segs"][int(full["seg_off"][i]):int(full["seg_off"][i + 1])]
                o = int(got["seg_off"][j])
                b = got["segs"][o:o + 2 * int(got["n_seeds"][j]) + 1]
                assert np.array_equal(a, b)
            ex = ctx.scan(extra)
            for j in range(2):
                o = int(got["extra_seg_off"][j])
                b = got["segs"][o:o + 2 * int(got["extra_n_seeds"][j]) + 1]
                assert np.array_equal(b, ex["segs"][int(ex["seg_off"][j]):int(ex["seg_off"][j + 1])])


def _rc_kmer(km, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (km & 3))
        km >>= 2
    return r


@pytest.mark.parametrize("k,L,e", [(13, 10000, 0.0), (10, 3000, 0.02)])
def test_select_seeds_matches_addseeds(ctx, k, L, e):
    """dp_select_seeds (A9 selection on the device) against the oracle's AddSeeds into an empty index."""
    N = 120
    bases, off = O.gen_reads(77 + k, N * L // 10, N, L, e, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    values = rs.kmer_values(k)
    ctx.upload_reads(bases, off)
    ctx.values_upload(values)
    wins, seqs = [], []
    rng = np.random.default_rng(5)
    for r in range(0, N, 2):
        Lr = int(off[r + 1] - off[r])
        text = bases[off[r]:off[r + 1]].tobytes().decode()
        view = O.Seq(text).sub(0, Lr)  # the cached view a later pass is served (seqio.go:115)
        for st, ln in ((0, min(1000, Lr)), (max(0, Lr - 1000), min(1000, Lr)), (int(rng.integers(0, Lr - 60)), 0)):
            if ln == 0:
                ln = int(rng.integers(2 * k - 1, min(Lr - st, 2600)))  # incl. windows shorter than one block and > 64 blocks
            wins.append((r, st, ln))
            seqs.append(view if (st == 0 and ln == Lr) else view.sub(st, st + ln))
    for num_seeds in (15, 7):
        top = ctx.select_seeds(wins, k, num_seeds)
        want = O.add_seeds_each(seqs, k, num_seeds, values)
        for i in range(len(wins)):
            got, seen = [], set()
            for km in top[i].tolist():  # AddSeeds :130-154: every list slot (incl. untouched zeros), then its RC
                for x in (km, _rc_kmer(km, k)):
                    if x not in seen:
                        seen.add(x)
                        got.append(x)
            assert got == want[i].tolist(), (num_seeds, wins[i])


def _index_from_sets(ctx, k, member):
    """member[s][i] truthy <=> sequence i holds seed s.  Builds the device index for it; returns the oracle IntSets."""
    S, M = len(member), len(member[0])
    ctx.round_begin(k, np.arange(S, dtype=np.uint32) * 37 + 5)  # any S distinct k-mers
    segs, offs, nseeds = [], [0], []
    for i in range(M):
        mine = [s for s in range(S) if member[s][i]]
        row = [3]
        for s in mine:
            row += [s, 2]
        segs += row
        offs.append(len(segs))
        nseeds.append(len(mine))
    ctx.import_segments(np.array(segs, dtype=np.int32))
    ctx.index_build(np.array(offs[:-1], dtype=np.uint64), np.array(nseeds, dtype=np.uint32))
    sets = [O.IntSet() for _ in range(S)]
    for s in range(S):
        for i in reversed(range(M)):  # IndexSequences adds in descending sequence order (seeds.go:373-381)
            if member[s][i]:
                sets[s].add(i)
    return sets


def _all_seed_query(S):
    q = [1]
    for s in range(S):
        q += [s, 1]
    return np.array(q, dtype=np.int32), np.array([0, len(q)], dtype=np.uint64)


def test_matches_reference_test2sharedids_vectors(ctx):
    """The reference's own GetSharedIDs known answers (util/bitset_test.go:38-161: 20 sets over 500 ids with
    multiplicities 16/8/4/2, minCount 16, 15, 8, 4, 2) driven through dp_index_build + dp_find_overlaps."""
    k = 10
    reads = ["ACGT" * 30]
    ctx.upload_reads(np.frombuffer(reads[0].encode(), dtype=np.uint8), np.array([0, len(reads[0])], dtype=np.int64))
    counts = np.zeros(500, dtype=np.int64)
    for i in range(500):
        counts[i] = 16 if i % 7 == 0 else 8 if i % 5 == 0 else 4 if i % 3 == 0 else 2 if i % 2 == 0 else 0
    member = [[j < counts[i] for i in range(500)] for j in range(20)]
    _index_from_sets(ctx, k, member)
    qs, qo = _all_seed_query(20)
    for hf, min_count, expect in ((0.8, 16, 16), (0.75, 15, 16), (0.4, 8, 8), (0.2, 4, 4), (0.1, 2, 2)):
        assert int(hf * 20 + 0.5) == min_count
        out = ctx.find_overlaps(qs, qo, hf, k, 500, want_candidates=True)
        want = [i for i in range(500) if counts[i] >= expect]
        assert out["cand"].tolist() == want, (min_count,)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_matches_all_ladder_regimes_vs_oracle(ctx, seed):
    """Matches -> GetSharedIDs(fast=true) for every threshold regime (<=1, 2-8 exact, 9-12 -> 8, 13-16 exact incl. the
    16-ladder's step-8 defect, 17-24 -> 16, > 24 exact-validated) against the oracle, sets in query-seed order."""
    k = 10
    ctx.upload_reads(np.frombuffer(b"ACGT" * 30, dtype=np.uint8), np.array([0, 120], dtype=np.int64))
    rng = np.random.default_rng(seed)
    S, M = 48, 700
    dens = rng.choice([0.05, 0.2, 0.45, 0.7], size=M)  # sequences with very different numbers of the query's seeds
    member = [[bool(rng.random() < dens[i]) for i in range(M)] for _ in range(S)]
    for s in range(S):  # uneven windows: some sets live in a narrow id range (early-return / swap-removal paths)
        if s % 5 == 0:
            lo = int(rng.integers(0, M - 150))
            member[s] = [member[s][i] and lo <= i < lo + 150 for i in range(M)]
    sets = _index_from_sets(ctx, k, member)
    qs, qo = _all_seed_query(S)
    for hf in (0.01, 0.03, 0.1, 0.16, 0.2, 0.24, 0.27, 0.32, 0.36, 0.41, 0.5, 0.52, 0.62, 0.8):
        min_count = int(hf * S + 0.5)
        out = ctx.find_overlaps(qs, qo, hf, k, 500, want_candidates=True)
        want = O.shared_ids(sets, min_count, True)
        assert out["cand"].tolist() == [int(x) for x in want], (seed, hf, min_count)


@pytest.mark.parametrize("S", [300, 511, 513, 900])
def test_matches_exact_count_regime_with_long_queries(ctx, S):
    """GetSharedIDs with minCount > 24 validates every bit of the 16-ladder's union by counting the sets that hold it
    (addSoftUnionIDs, util/bitset.go:509-538).  A query of 300 - 511 usable seeds (`overlap` with -overlap_size 2000 -num_seeds 30 on
    a small genome) has sequences holding more than 255 of them: the device's bit-sliced counter had eight planes until round 5 and
    dropped exactly those - the best candidates.  Sets in query-seed order, thresholds on both sides of 255."""
    k = 10
    ctx.upload_reads(np.frombuffer(b"ACGT" * 30, dtype=np.uint8), np.array([0, 120], dtype=np.int64))
    rng = np.random.default_rng(S)
    M = 400
    dens = rng.choice([0.02, 0.3, 0.6, 0.85, 0.97, 1.0], size=M)
    member = [[bool(rng.random() < dens[i]) for i in range(M)] for _ in range(S)]
    assert max(sum(member[s][i] for s in range(S)) for i in range(M)) > 255
    sets = _index_from_sets(ctx, k, member)
    qs, qo = _all_seed_query(S)
    # (S > 512: more sets than query_kernel's LDS lists hold - the BIG variants, lists in global memory; small fractions take their
    # 4- / 8-ladder, 0.012 * 900 = 11 the 8-ladder's saturation, 0.016 * 900 = 14 the 16-ladder with its step-8 omission)
    for hf in (0.005, 0.012, 0.016, 0.09, 0.4, 0.8, 0.86, 0.99):
        min_count = int(hf * S + 0.5)
        out = ctx.find_overlaps(qs, qo, hf, k, 2 * S + 8, want_candidates=True)
        want = [int(x) for x in O.shared_ids(sets, min_count, True)]
        assert len(want) > 0
        assert out["cand"].tolist() == want, (S, hf, min_count)


def test_reference_sequence_test_vectors_on_device(ctx):
    """sequence/sequence_test.go on the device: packBytes("CGGT") = 0x6B (Test9Packing), CountKmers = 27 for k=6 on the
    70-base test sequence with the test's kmerSet (Test6CountKmers), the k=8/k=11 counts on SubSequence(7, len-7), and
    WriteSegments on the full sequence and on SubSequence(2, len-2) (Test8Segments)."""
    from tests.test_oracle_known_answers import S70, kmer_set
    reads = [S70, "CGGT" * 5]
    bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    off = np.array([0, len(S70), len(S70) + 20], dtype=np.int64)
    ctx.upload_reads(bases, off)
    assert bytes(ctx.packed_read(1)) == bytes([0x6B] * 5)

    def seeds_of(ks):
        return np.nonzero(ks)[0].astype(np.uint32)

    def scan_view(start, end, k, seeds):
        ctx.round_begin(k, seeds)
        res = ctx.scan([(0, start, (end - start) - k + 1, 0)])
        segs = res["segs"][int(res["seg_off"][0]):int(res["seg_off"][1])]
        # device seed ids are positions in `seeds`; the reference's arrays hold k-mers at this level
        out = segs.astype(np.int64).copy()
        out[1::2] = seeds[segs[1::2]]
        return int(res["n_seeds"][0]), out

    ks, count = kmer_set(S70, 6)
    n, segs = scan_view(0, len(S70), 6, seeds_of(ks))
    assert n == count == 27
    assert np.array_equal(segs, O.byte_write_segments(S70, 6, ks))
    n2, segs2 = scan_view(2, len(S70) - 2, 6, seeds_of(ks))
    assert np.array_equal(segs2, O.byte_write_segments(S70[2:len(S70) - 2], 6, ks))
    sub = S70[7:len(S70) - 7]
    for k in (8, 11):
        ksk, cnt = kmer_set(sub, k)
        n, _ = scan_view(7, len(S70) - 7, k, seeds_of(ksk))
        assert n == cnt == O.byte_count_kmers(sub, 100, k, ksk)
