"""Known answers ABOVE the primitives, worked by hand from the reference's Go text (tests/golden/hand/: every file carries its
derivation with file:line citations) - the only vectors at this level that do not come out of this repository's own code.

CPU (`-m "not gpu"`): the oracle against them, and the product's host-side rules (libdownpore_host.so: isConsistent, removeDominated,
trimToBestSeed's first step - C++ written independently of the oracle's).  GPU (`-m gpu`): the product's device path - the chaining
kernels behind dp_find_overlaps, chunk_kernel behind dp_index_build_chunked, map_kernel behind dp_map_windows - on the same cases.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as O

HAND = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hand")
i64p = C.POINTER(C.c_int64)


def hand(name):
    return json.load(open(os.path.join(HAND, name)))


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(i64p)


def _oracle():
    L = O.lib()
    L.dpo_hand_chunks.argtypes = [i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, i64p, C.c_int64, i64p]
    L.dpo_hand_trim_indices.argtypes = [C.c_int64, i64p, i64p, C.c_int64, C.c_int64, C.c_int64, i64p]
    L.dpo_hand_is_consistent.argtypes = [i64p, i64p, C.c_int, C.c_int64]
    L.dpo_hand_remove_dominated.argtypes = [i64p, C.c_int, C.c_int64, C.POINTER(C.c_int)]
    return L


def _host():
    from downpore_amd.overlap import load_host
    H = load_host()
    H.dph_hand_is_consistent.argtypes = [i64p, i64p, C.c_int, C.c_int64]
    H.dph_hand_remove_dominated.argtypes = [i64p, C.c_int, C.c_int64, C.POINTER(C.c_int)]
    H.dph_hand_trim_indices.restype = None
    H.dph_hand_trim_indices.argtypes = [C.c_int, C.POINTER(C.c_int32), i64p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    return H


# ---------------------------------------------------------------------------------------------------------------- oracle (CPU)
def test_gap_range_oracle():
    for c in hand("gap_range.json")["cases"]:
        out = np.zeros(2, dtype=np.int64)
        O.lib().dpo_gap_range(c["gap"], c["k"], _p(out))
        assert out.tolist() == [c["min"], c["max"]], c


@pytest.mark.parametrize("name", ["pairwise_two_chains.json", "pairwise_break_after_first_extension.json"])
def test_pairwise_oracle(name):
    h = hand(name)
    got = O.pairwise(h["a_segments"], h["b_segments"], h["min_matches"], h["k"])
    assert [(a.tolist(), b.tolist()) for a, b in got] == [(m["match_a"], m["match_b"]) for m in h["expect_pairwise"]]


def _chunk_segments(h):
    r = h["read"]
    seg = [r["first_gap"]]
    for i in range(r["seeds"]):
        seg += [1000 + i, r["gap"]]
    seg[-1] = r["last_gap"]
    assert seg[0] + r["seeds"] * h["k"] + (r["seeds"] - 1) * r["gap"] + r["last_gap"] == r["length"]
    return seg


def test_chunk_worker_oracle():
    h = hand("chunk_three_pieces.json")
    seg = _i64(_chunk_segments(h))
    out = np.zeros(5 * 16, dtype=np.int64)
    n = C.c_int64(0)
    assert _oracle().dpo_hand_chunks(_p(seg), len(seg), h["read"]["length"], 0, 0, h["chunk_size"], h["overlap"], h["min_seeds"], h["k"],
                                     _p(out), 16, C.byref(n)) == 0
    got = out[:5 * n.value].reshape(-1, 5).tolist()
    assert got == [[e["first_seed"], e["seeds"], e["length"], e["offset"], e["inset"]] for e in h["expect"]]


def _trim_args(c, dtype):
    flat = np.ascontiguousarray(sum(c["match_a"], []), dtype=dtype)
    off = _i64(np.cumsum([0] + [len(m) for m in c["match_a"]]))
    return flat, off


def test_trim_indices_oracle():
    for c in hand("trim_indices.json")["cases"]:
        flat, off = _trim_args(c, np.int64)
        out = np.zeros(2, dtype=np.int64)
        assert _oracle().dpo_hand_trim_indices(c["upto"], _p(flat), _p(off), len(c["match_a"]), c["min_match"], c["length"], _p(out)) == 0
        assert out.tolist() == [c["best"], c["back"]], c["why"]


def test_is_consistent_oracle():
    L = _oracle()
    for c in hand("is_consistent.json")["cases"]:
        got = L.dpo_hand_is_consistent(_p(_i64(c["left"])), _p(_i64(c["right"])), c["circular"], c["ref_len"])
        assert bool(got) == c["want"], (c["distance"], c["expected_distance"], c["why"])


def test_remove_dominated_oracle():
    L = _oracle()
    h = hand("remove_dominated.json")
    for c in h["cases"]:
        maps = _i64(c["maps"]).reshape(-1)
        kept = (C.c_int * len(c["maps"]))()
        n = L.dpo_hand_remove_dominated(_p(maps), len(c["maps"]), h["query_len"], kept)
        assert list(kept[:n]) == c["kept"], c["why"]


def test_extend_chain_skip_oracle():
    h = hand("extend_chain_skip.json")
    got = O.match(h["target_segments"], h["query_segments"], h["min_match"], h["k"])
    assert [(a.tolist(), b.tolist()) for a, b in got] == [(m["match_a"], m["match_b"]) for m in h["expect"]]


# ------------------------------------------------------------------------------------------- product, host-side rules (CPU: no kernel runs)
def test_is_consistent_product_host():
    H = _host()
    for c in hand("is_consistent.json")["cases"]:
        got = H.dph_hand_is_consistent(_p(_i64(c["left"])), _p(_i64(c["right"])), c["circular"], c["ref_len"])
        assert bool(got) == c["want"], (c["distance"], c["expected_distance"], c["why"])


def test_remove_dominated_product_host():
    H = _host()
    h = hand("remove_dominated.json")
    for c in h["cases"]:
        maps = _i64(c["maps"]).reshape(-1)
        kept = (C.c_int * len(c["maps"]))()
        n = H.dph_hand_remove_dominated(_p(maps), len(c["maps"]), h["query_len"], kept)
        assert list(kept[:n]) == c["kept"], c["why"]


def test_trim_indices_product_host():
    H = _host()
    for c in hand("trim_indices.json")["cases"]:
        flat, off = _trim_args(c, np.int32)
        out = (C.c_int * 2)()
        H.dph_hand_trim_indices(c["upto"], flat.ctypes.data_as(C.POINTER(C.c_int32)), _p(off), len(c["match_a"]), c["min_match"], c["length"], out)
        assert [out[0], out[1]] == [c["best"], c["back"]], c["why"]


# ------------------------------------------------------------------------------------------------------- product, device path (GPU)
@pytest.fixture()
def ctx():
    from downpore_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def _device_index(ctx, k, seqs, n_seeds=64):
    """seqs: segment arrays; the device index of exactly these sequences (seed ids < n_seeds)."""
    ctx.upload_reads(np.frombuffer(b"ACGT" * 30, dtype=np.uint8), np.array([0, 120], dtype=np.int64))
    ctx.round_begin(k, np.arange(n_seeds, dtype=np.uint32) * 37 + 5)  # any n_seeds distinct k-mers: the kernels see seed ids only
    flat, offs, ns = [], [], []
    for s in seqs:
        offs.append(len(flat))
        ns.append(len(s) // 2)
        flat += s
    ctx.import_segments(np.array(flat, dtype=np.int32))
    ctx.index_build(np.array(offs, dtype=np.uint64), np.array(ns, dtype=np.uint32))


def _add_seeds_case():
    """-> (k, text, [(values, {minSeeds: expected seed k-mers})]): the case and its variant with a tie between blocks"""
    h = hand("add_seeds_blocks.json")
    k, text = h["k"], h["sequence"]
    runs = []
    for over, expect in (({}, h["expect"]), (h["tie"]["values_override"], h["tie"]["expect"])):
        values = np.zeros(4 ** k, dtype=np.float64)
        for st, v in list(h["values_at_start"].items()) + list(over.items()):
            values[O.kmer_value(text[int(st):int(st) + k])] = v
        runs.append((values, expect))
    return k, text, runs


def test_add_seeds_oracle():
    """AddSeeds of a 100-base sequence into an empty index: blocks, bests, the top-N list, the order of the seeds - worked by hand."""
    k, text, runs = _add_seeds_case()
    for values, expect in runs:
        for n, want in expect.items():
            got = O.add_seeds_each([O.Seq(text)], k, int(n), values)[0].tolist()
            assert got == [O.kmer_value(x) for x in want], n


def test_add_seeds_host():
    k, text, runs = _add_seeds_case()
    H = _host()
    H.dph_hand_add_seeds.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    for values, expect in runs:
        for n, want in expect.items():
            out = np.zeros(64, dtype=np.uint32)
            m = H.dph_hand_add_seeds(text.encode(), len(text), k, int(n), values.ctypes.data, out.ctypes.data, 64)
            assert out[:m].tolist() == [O.kmer_value(x) for x in want], n


def test_bases_covered_oracle_and_host():
    """GetBasesCovered - the PAF line's tenth column - on overlapping seeds, a skipped seed, a single seed."""
    h = hand("bases_covered.json")
    a, b = np.array(h["a_segments"], dtype=np.int64), np.array(h["b_segments"], dtype=np.int64)
    a32, b32 = a.astype(np.int32), b.astype(np.int32)
    L, H = O.lib(), _host()
    L.dpo_hand_bases_covered.argtypes = [i64p, C.c_int64, i64p, C.c_int64, i64p, i64p, C.c_int64, C.c_int, i64p]
    i32p = C.POINTER(C.c_int32)
    H.dph_hand_bases_covered.argtypes = [i32p, C.c_int, i32p, C.c_int, i32p, i32p, C.c_int, C.c_int, i64p]
    for c in h["cases"]:
        ma, mb = np.array(c["match_a"], dtype=np.int64), np.array(c["match_b"], dtype=np.int64)
        out = np.zeros(2, dtype=np.int64)
        assert L.dpo_hand_bases_covered(_p(a), len(a), _p(b), len(b), _p(ma), _p(mb), len(ma), h["k"], _p(out)) == 0
        assert out.tolist() == [c["count_a"], c["count_b"]], ("oracle", c)
        ma32, mb32 = ma.astype(np.int32), mb.astype(np.int32)
        out[:] = 0
        assert H.dph_hand_bases_covered(a32.ctypes.data_as(i32p), len(a32), b32.ctypes.data_as(i32p), len(b32), ma32.ctypes.data_as(i32p),
                                        mb32.ctypes.data_as(i32p), len(ma32), h["k"], _p(out)) == 0
        assert out.tolist() == [c["count_a"], c["count_b"]], ("host", c)


CONSENSUS_CASES = ["consensus_three_sequences.json", "consensus_third_sequence_skips_a_seed.json"]


def _consensus_case(name):
    h = hand(name)
    names = list(h["sequences"].keys())
    segs = [v for n in names for v in h["sequences"][n]]
    off = np.cumsum([0] + [len(h["sequences"][n]) for n in names]).astype(np.int64)
    return h, names, segs, off


def _check_consensus(h, names, cons, kept, counts, a, b):
    e = h["expect"]
    assert cons == e["consensus"]
    assert [names[i] for i in kept] == e["kept"]
    at = 0
    for j, i in enumerate(kept):
        n = counts[j]
        assert a[at:at + n] == e["match_a"][names[i]] and b[at:at + n] == e["match_b"][names[i]], names[i]
        at += n


@pytest.mark.parametrize("name", CONSENSUS_CASES)
def test_consensus_oracle(name):
    """multiAligner.Consensus of three sequences, worked by hand step by step (the files' derivations)."""
    h, names, segs, off = _consensus_case(name)
    L = O.lib()
    L.dpo_hand_consensus.argtypes = [i64p, i64p, C.c_int, C.c_int, i64p, C.c_int64, i64p, C.POINTER(C.c_int), i64p, i64p, i64p, C.c_int64, i64p]
    sg = np.array(segs, dtype=np.int64)
    cons, ncons = np.zeros(64, dtype=np.int64), np.zeros(1, dtype=np.int64)
    kept = (C.c_int * 8)()
    counts, a, b, nm = np.zeros(8, dtype=np.int64), np.zeros(64, dtype=np.int64), np.zeros(64, dtype=np.int64), np.zeros(1, dtype=np.int64)
    assert L.dpo_hand_consensus(_p(sg), _p(off), len(names), h["k"], _p(cons), 64, _p(ncons), kept, _p(counts), _p(a), _p(b), 64, _p(nm)) == 0
    n = int(nm[0])
    _check_consensus(h, names, cons[:int(ncons[0])].tolist(), [kept[j] for j in range(n)], counts[:n].tolist(), a.tolist(), b.tolist())


@pytest.mark.parametrize("name", CONSENSUS_CASES)
def test_consensus_host(name):
    """The same cases through the product's host consensus (the path of windows no device layout holds)."""
    h, names, segs, off = _consensus_case(name)
    H = _host()
    i32p = C.POINTER(C.c_int32)
    H.dph_hand_consensus.argtypes = [i32p, i64p, C.c_int, C.c_int, i32p, C.c_int64, i64p, C.POINTER(C.c_int), i64p, i32p, i32p, C.c_int64, i64p]
    sg = np.array(segs, dtype=np.int32)
    cons, ncons = np.zeros(64, dtype=np.int32), np.zeros(1, dtype=np.int64)
    kept = (C.c_int * 8)()
    counts, a, b, nm = np.zeros(8, dtype=np.int64), np.zeros(64, dtype=np.int32), np.zeros(64, dtype=np.int32), np.zeros(1, dtype=np.int64)
    rc = H.dph_hand_consensus(sg.ctypes.data_as(i32p), _p(off), len(names), h["k"], cons.ctypes.data_as(i32p), 64, _p(ncons), kept, _p(counts),
                              a.ctypes.data_as(i32p), b.ctypes.data_as(i32p), 64, _p(nm))
    assert rc == 0
    n = int(nm[0])
    _check_consensus(h, names, cons[:int(ncons[0])].tolist(), [kept[j] for j in range(n)], counts[:n].tolist(), a.tolist(), b.tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pairwise_two_chains.json", "pairwise_break_after_first_extension.json"])
def test_pairwise_device(ctx, name):
    h = hand(name)
    _device_index(ctx, h["k"], [h["b_segments"]] + h["decoy_segments"])
    q = np.array(h["a_segments"], dtype=np.int32)
    out = ctx.find_overlaps(q, np.array([0, len(q)], dtype=np.uint64), 0.25, h["k"], 500, want_candidates=True)
    assert out["cand"].tolist() == h["expect_device"]["candidates"]
    got = [(int(out["target"][i]), out["match_a"][int(out["off"][i]):int(out["off"][i + 1])].tolist(),
            out["match_b"][int(out["off"][i]):int(out["off"][i + 1])].tolist()) for i in range(len(out["target"]))]
    assert got == [(m["target"], m["match_a"], m["match_b"]) for m in h["expect_device"]["matches"]]


def _enc(kmer):
    v = 0
    for b in kmer:
        v = (v << 2) | {65: 0, 67: 1, 71: 2, 84: 3}[b]
    return v


@pytest.mark.gpu
def test_chunk_worker_device(ctx):
    """chunk_kernel on a read whose 250 seed occurrences sit exactly where the hand case puts them: a random read, the round's seeds =
    the 13-mers at bases 7 + 43 i (checked here to occur nowhere else in the read), scanned on the device, chunked by
    dp_index_build_chunked."""
    h = hand("chunk_three_pieces.json")
    k, r = h["k"], h["read"]
    pos = [r["first_gap"] + (k + r["gap"]) * i for i in range(r["seeds"])]
    for seed in range(1, 50):
        rng = np.random.default_rng(seed)
        read = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=r["length"]))
        kmers = [read[p:p + k] for p in pos]
        where = {}
        for p in range(r["length"] - k + 1):
            where.setdefault(read[p:p + k], []).append(p)
        if len(set(kmers)) == len(kmers) and all(where[km] == [p] for km, p in zip(kmers, pos)):
            break
    else:
        raise AssertionError("no read without chance hits found")
    ctx.upload_reads(np.frombuffer(read, dtype=np.uint8), np.array([0, len(read)], dtype=np.int64))
    ctx.round_begin(k, np.array([_enc(km) for km in kmers], dtype=np.uint32))
    sc = ctx.scan_reads(np.zeros(1, dtype=np.uint8), 1, 0, 1, False, h["min_seeds"])
    assert sc["read"].tolist() == [0] and sc["n_seeds"].tolist() == [r["seeds"]]
    L = ctx.L
    L.dp_index_build_chunked.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32)]
    cap = C.c_uint32(0)
    ctx._chk(L.dp_index_build_chunked(ctx.h, h["chunk_size"], h["overlap"], h["min_seeds"], 0, 1, C.byref(cap)))
    refs = np.zeros(cap.value + 4, dtype=[("seg_off", np.uint64), ("n_seeds", np.uint32), ("reserved", np.uint32)])
    metas = np.zeros(cap.value + 4, dtype=[("read", np.uint32), ("length", np.int32), ("offset", np.int32), ("inset", np.int32)])
    L.dp_index_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    n = C.c_uint32(0)
    ctx._chk(L.dp_index_chunks(ctx.h, refs.ctypes.data, metas.ctypes.data, len(refs), C.byref(n)))
    base = int(sc["seg_off"][0])
    got = [[(int(refs["seg_off"][i]) - base) // 2, int(refs["n_seeds"][i]), int(metas["length"][i]), int(metas["offset"][i]), int(metas["inset"][i])]
           for i in range(n.value)]
    assert got == [[e["first_seed"], e["seeds"], e["length"], e["offset"], e["inset"]] for e in h["expect"]]


@pytest.mark.gpu
def test_extend_chain_skip_device(ctx):
    h = hand("extend_chain_skip.json")
    _device_index(ctx, h["k"], [h["target_segments"]] + h["decoy_segments"])
    fwd = h["query_segments"]
    rc = [h["window_len"]]  # the reverse-complement window holds none of the round's seeds: one gap
    segs = np.array(fwd + rc, dtype=np.int32)
    off = np.array([0, len(fwd), len(fwd) + len(rc)], dtype=np.uint64)
    out = ctx.map_windows(segs, off, np.array([h["window_len"], h["window_len"]], dtype=np.uint32), h["k"])
    got = [(int(out["window"][i]), int(out["target"][i]), out["match_a"][int(out["off"][i]):int(out["off"][i + 1])].tolist(),
            out["match_b"][int(out["off"][i]):int(out["off"][i + 1])].tolist()) for i in range(len(out["window"]))]
    assert got == [(0, 0, m["match_a"], m["match_b"]) for m in h["expect"]]


class _ConsBatch(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("cons", C.POINTER(C.c_int32)), ("cons_off", C.POINTER(C.c_uint64)), ("cons_len", C.POINTER(C.c_uint32)),
                ("match_a", C.POINTER(C.c_int32)), ("match_b", C.POINTER(C.c_int32)), ("match_len", C.POINTER(C.c_uint32)),
                ("flags", C.POINTER(C.c_uint32)), ("kernel_ms", C.c_double)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONSENSUS_CASES)
def test_consensus_device(ctx, name):
    """The same cases on the device's alignment kernel (dp_consensus_align: it takes the Reduced() sequences - here the sequences
    themselves, every seed being shared - and returns every sequence's pairs; the caller drops the ones with fewer than three)."""
    h, names, segs, off = _consensus_case(name)
    sg = np.array(segs, dtype=np.int32)
    so = off.astype(np.uint64)
    go = np.array([0, len(names)], dtype=np.uint32)
    out = _ConsBatch()
    L = ctx.L
    L.dp_consensus_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(_ConsBatch)]
    assert L.dp_consensus_align(ctx.h, sg.ctypes.data, so.ctypes.data, go.ctypes.data, 1, h["k"], C.byref(out)) == 0
    assert out.n_groups == 1 and out.flags[0] == 0
    cons = [out.cons[int(out.cons_off[0]) + i] for i in range(int(out.cons_len[0]))]
    kept, counts, a, b = [], [], [], []
    for i in range(len(names)):
        n = int(out.match_len[i])
        if n >= 3:   # seeds/alignment.go:258-266
            kept.append(i)
            counts.append(n)
            a += [out.match_a[int(so[i]) + x] for x in range(n)]
            b += [out.match_b[int(so[i]) + x] for x in range(n)]
    if "C" not in h["expect"]["kept"]:
        assert int(out.match_len[names.index("C")]) == 0
    _check_consensus(h, names, cons, kept, counts, a, b)


@pytest.mark.gpu
def test_add_seeds_device(ctx):
    """The selection kernel (dp_select_seeds) returns the top-N list in slot order; AddSeeds then adds every slot and its reverse
    complement (seeds.go:130-154) - the hand case's seed order."""
    k, text, runs = _add_seeds_case()
    b = np.frombuffer(text.encode(), dtype=np.uint8)
    ctx.upload_reads(b, np.array([0, len(b)], dtype=np.int64))
    for values, expect in runs:
        ctx.values_upload(values)
        for n, want in expect.items():
            top = ctx.select_seeds([(0, 0, len(text))], k, int(n))[0].tolist()
            got, seen = [], set()
            for km in top:
                r, x = 0, km
                for _ in range(k):
                    r = (r << 2) | (3 - (x & 3))
                    x >>= 2
                for y in (km, r):
                    if y not in seen:
                        seen.add(y)
                        got.append(y)
            assert got == [O.kmer_value(x) for x in want], n
