"""Pins the ORACLE against every known-answer vector the reference's own unit tests hold
(sequence/sequence_test.go, util/bitset_test.go) plus the asm-level quirks SURVEY.md §8(a) lists.
CPU only.
"""
import numpy as np
import pytest

from tests import oracle_lib as O

S70 = "GGGAAGTGACTGCCTTAAAATGAGGGTTACCCCTTTTAGTTGACAAGACGCTTGCGGCTATTATGGCTAG"  # sequence_test.go:7-9
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def kmer_set(s, k):
    """sequence_test.go:11-40"""
    ks = np.zeros(4 ** k, dtype=np.uint8)
    count = 0
    for i in range(4, len(s) - k - 1, 5):
        ks[O.kmer_value(s[i:i + k])] = 1
        ks[O.kmer_value(s[i + 1:i + 1 + k])] = 1
        count += 2
    for sub in (s[0:k], s[1:k + 1], s[len(s) - k:]):
        x = O.kmer_value(sub)
        if not ks[x]:
            ks[x] = 1
            count += 1
    return ks, count


def test_lengths_and_strings():  # Test1Lengths, Test2String
    for i in range(0, 5):
        s = S70[:len(S70) - i]
        p = O.Seq(s)
        assert len(p) == len(s)
        assert str(p) == s


def test_reverse_complement():  # Test3Complement
    assert str(O.Seq(S70).rc()) == revcomp(S70)


def test_subsequences():  # Test4Subs
    p = O.Seq(S70)
    for i in range(15, 20):
        assert str(p.sub(i - 15, i)) == S70[i - 15:i]
        assert str(p.sub(i, i + 30)) == S70[i:i + 30]


def test_kmer_at_and_next():  # Test5KmerAt, Test7IterateKmers
    p = O.Seq(S70)
    mask = 4 ** 6 - 1
    for i in range(0, len(S70) - 6):
        a = p.kmer_at(i, 6)
        assert a == O.kmer_value(S70[i:i + 6])
        assert p.next_kmer(a, mask, i + 6) == O.kmer_value(S70[i + 1:i + 7])


def test_count_kmers():  # Test6CountKmers
    ks, count = kmer_set(S70, 6)
    p = O.Seq(S70)
    c1 = O.byte_count_kmers(S70, 100, 6, ks)
    c2 = p.count_kmers(100, 6, ks)
    assert c1 == c2 == count == 27  # 27 for k=6 on the full sequence
    assert O.byte_count_kmers(S70, 7, 6, ks) >= 7 and p.count_kmers(7, 6, ks) >= 7
    sub = S70[7:len(S70) - 7]
    ps = p.sub(7, len(S70) - 7)
    # the reference also runs k=17 (a 16 GiB table); k=11 exercises the same >8-base straddling path
    ks, count = kmer_set(sub, 8)
    assert O.byte_count_kmers(sub, 100, 8, ks) == ps.count_kmers(100, 8, ks) == count
    ks, count = kmer_set(sub, 11)
    assert O.byte_count_kmers(sub, 100, 11, ks) == ps.count_kmers(100, 11, ks) == count


def test_write_segments():  # Test8Segments
    ks, count = kmer_set(S70, 6)
    p = O.Seq(S70)
    assert np.array_equal(O.byte_write_segments(S70, 6, ks), p.write_segments(6, ks))
    sub = S70[2:len(S70) - 2]
    assert np.array_equal(O.byte_write_segments(sub, 6, ks), p.sub(2, len(S70) - 2).write_segments(6, ks))


def test_packing():  # Test9Packing
    import ctypes as C
    out = np.zeros(2, dtype=np.uint8)
    O.lib().dpo_pack_bytes(b"CGGT", 4, O.ptr(out, O.u8p))
    assert out[0] == 0x6B and out[1] == 0
    s = b"CGGT" * 5
    out = np.zeros(len(s) // 4 + 1, dtype=np.uint8)
    O.lib().dpo_pack_bytes(s, len(s), O.ptr(out, O.u8p))
    assert all(b == 0x6B for b in out[:-1]) and out[-1] == 0


def test_len_mod4_defect():
    """a top-level read with len%4==0 never examines its last 4 k-mers and its final
    gap is 4 short (len 68 => 24 of the 27 expected hits with the all-ones table restricted to S70's k-mers)."""
    s = S70[:68]
    ks = np.zeros(4 ** 6, dtype=np.uint8)
    for i in range(len(s) - 5):
        ks[O.kmer_value(s[i:i + 6])] = 1
    p = O.Seq(s)  # top level: finalLen = 0
    assert p.meta()["finalLen"] == 0
    full = O.byte_count_kmers(s, 1000, 6, ks)
    assert full == len(s) - 5
    assert p.count_kmers(1000, 6, ks) == full - 4
    seg = p.write_segments(6, ks)
    assert seg[-1] + (len(seg) // 2) * 6 + seg[:-1:2].sum() == len(s) - 4
    # the cached-read view (seqio.go:115) is not affected
    v = p.sub(0, len(s))
    assert v.meta()["finalLen"] == 4 and v.meta()["inset"] == 1  # inset off-by-one of SubSequence (:365)
    assert v.count_kmers(1000, 6, ks) == full
    assert np.array_equal(v.write_segments(6, ks), O.byte_write_segments(s, 6, ks))


def test_segments_invariant_random():
    rng = np.random.default_rng(5)
    for trial in range(50):
        L = int(rng.integers(40, 400))
        s = "".join("ACGT"[x] for x in rng.integers(0, 4, L))
        k = int(rng.integers(4, 12))
        ks = (rng.random(4 ** k) < 0.2).astype(np.uint8)
        p = O.Seq(s)
        a = int(rng.integers(0, 8))
        b = L - int(rng.integers(0, 8))
        v = p.sub(a, b)
        ref = O.byte_write_segments(s[a:b], k, ks)
        got = v.write_segments(k, ks)
        assert np.array_equal(ref, got), (trial, L, k, a, b)
        assert v.count_kmers(10 ** 9, k, ks) == len(ref) // 2
        assert got[::2].sum() + (len(got) // 2) * k == b - a


# ---- util/bitset_test.go ------------------------------------------------------------------------

def test_count_intersection():  # Test1CountIntersection
    a, b = O.IntSet(), O.IntSet()
    for i in range(1001, 3000, 5):
        a.add(i)
    count = 0
    for j in range(101, 2013, 3):
        b.add(j)
        if a.contains(j):
            count += 1
    assert a.count_intersection(b) == count == b.count_intersection(a)
    assert b.count_intersection_to(a, count + 10) == count
    assert a.count_intersection_to(b, count + 10) == count
    # early exit happens only between 8-word blocks: result >= maxCount, never below the true count cap
    assert a.count_intersection_to(b, 3) >= 3


@pytest.mark.parametrize("fast", [False, True])
def test_shared_ids(fast):  # Test2SharedIDs
    sets = [O.IntSet() for _ in range(20)]
    counts = np.zeros(500, dtype=np.int64)
    for i in range(500):
        if i % 7 == 0:
            counts[i] = 16
        elif i % 5 == 0:
            counts[i] = 8
        elif i % 3 == 0:
            counts[i] = 4
        elif i % 2 == 0:
            counts[i] = 2
        for j in range(counts[i]):
            sets[j].add(i)
    for min_count, expect in ((16, 16), (15, 16), (8, 8), (4, 4), (2, 2)):
        ids = O.shared_ids(sets, min_count, fast)
        assert len(ids) == int((counts >= expect).sum())
        assert all(counts[i] >= expect for i in ids)
        assert np.all(np.diff(ids.astype(np.int64)) > 0)  # ascending


def test_soft_union_ladders():
    rng = np.random.default_rng(3)
    for n in (1, 3, 4, 6, 9, 13, 20):
        w = rng.integers(0, 2 ** 63, n, dtype=np.uint64) & rng.integers(0, 2 ** 63, n, dtype=np.uint64)
        bits = np.array([[(int(x) >> b) & 1 for b in range(64)] for x in w])
        tot = bits.sum(axis=0)

        def mask(th):
            return sum(1 << b for b in range(64) if tot[b] >= th)

        v = O.soft_union(4, w)
        assert [int(x) for x in v] == [mask(1), mask(2), mask(3), mask(4)]
        if n >= 6:
            v = O.soft_union(8, w)
            assert [int(x) for x in v] == [mask(5), mask(6), mask(7), mask(8)]
        if n >= 13:
            # step-8 defect: a bit whose first occurrence is in the 8th word is under-counted by one
            eff = tot.copy()
            for b in range(64):
                first = int(np.argmax(bits[:, b])) if tot[b] else -1
                if first == 7:
                    eff[b] -= 1
            v = O.soft_union(16, w)
            want = [sum(1 << b for b in range(64) if eff[b] >= th) for th in (13, 14, 15, 16)]
            assert [int(x) for x in v] == want


def test_intset_add_semantics():
    s = O.IntSet()
    assert s.window() == (1, 0, 50)
    s.add(5000)  # grows to index+2 words (bitset.go:78-82)
    assert s.window() == (78, 78, 80) and s.size() == 1
    s.add(3)
    assert s.window()[0] == 0 and s.size() == 2
    s.add(3)
    assert s.size() == 2 and s.contains(3) and not s.contains(4)


def test_gap_range():
    out = np.zeros(2, dtype=np.int64)
    for gap, k, want in ((0, 10, (-10, 11)), (-30, 10, (-10, 0)), (3, 2, (0, 20)), (300, 13, (187, 464)),
                         (3, 13, (-13, 18))):
        O.lib().dpo_gap_range(gap, k, O.ptr(out, O.i64p))
        assert tuple(int(x) for x in out) == want, (gap, k, tuple(out))
