"""ctypes front end to the ORACLE (oracle/_build/liboracle.so) and the synthetic-read generator.

Test infrastructure only: nothing under downpore_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# DPO_LIB: another build of the same checker (the AddressSanitizer build of `make -C oracle asan`, tests/test_host_asan.py)
_ORACLE_SO = os.environ.get("DPO_LIB") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")
_SYNTH_SO = os.path.join(ROOT, "tools", "libdpsynth.so")

i64p = C.POINTER(C.c_int64)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j4"])
    if not os.path.exists(_SYNTH_SO) or os.path.getmtime(_SYNTH_SO) < os.path.getmtime(os.path.join(ROOT, "tools", "synth.cpp")):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tools", "synth.cpp"),
                               "-o", _SYNTH_SO])


_lib = None
_synth = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_ORACLE_SO):
            build_oracle()
        L = C.CDLL(_ORACLE_SO)
        vp = C.c_void_p
        L.dpo_last_error.restype = C.c_char_p
        L.dpo_seq_new.restype = vp
        L.dpo_seq_new.argtypes = [C.c_char_p, C.c_int64]
        L.dpo_seq_free.argtypes = [vp]
        L.dpo_seq_sub.restype = vp
        L.dpo_seq_sub.argtypes = [vp, C.c_int64, C.c_int64]
        L.dpo_seq_rc.restype = vp
        L.dpo_seq_rc.argtypes = [vp]
        L.dpo_seq_str.restype = C.c_int64
        L.dpo_seq_str.argtypes = [vp, C.c_char_p, C.c_int64]
        L.dpo_seq_meta.argtypes = [vp, i64p]
        L.dpo_seq_bytes.argtypes = [vp, u8p]
        L.dpo_seq_kmer_at.restype = C.c_int64
        L.dpo_seq_kmer_at.argtypes = [vp, C.c_int64, C.c_int]
        L.dpo_seq_next_kmer.restype = C.c_int64
        L.dpo_seq_next_kmer.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64]
        L.dpo_seq_count_kmers.restype = C.c_int64
        L.dpo_seq_count_kmers.argtypes = [vp, C.c_int64, C.c_int, u8p]
        L.dpo_seq_count_kmers_between.restype = C.c_int64
        L.dpo_seq_count_kmers_between.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, u8p]
        L.dpo_seq_write_segments.restype = C.c_int64
        L.dpo_seq_write_segments.argtypes = [vp, C.c_int, u8p, i64p]
        L.dpo_byte_count_kmers.restype = C.c_int64
        L.dpo_byte_count_kmers.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int, u8p]
        L.dpo_byte_write_segments.restype = C.c_int64
        L.dpo_byte_write_segments.argtypes = [C.c_char_p, C.c_int64, C.c_int, u8p, i64p]
        L.dpo_pack_bytes.argtypes = [C.c_char_p, C.c_int64, u8p]
        L.dpo_rc_kmer.restype = C.c_uint64
        L.dpo_rc_kmer.argtypes = [C.c_uint64, C.c_int]
        L.dpo_set_new.restype = vp
        L.dpo_set_new_cap.restype = vp
        L.dpo_set_new_cap.argtypes = [C.c_int64]
        L.dpo_set_free.argtypes = [vp]
        L.dpo_set_add.argtypes = [vp, C.c_uint64]
        L.dpo_set_clear.argtypes = [vp]
        L.dpo_set_contains.argtypes = [vp, C.c_uint64]
        L.dpo_set_size.restype = C.c_uint64
        L.dpo_set_size.argtypes = [vp]
        L.dpo_set_window.argtypes = [vp, u64p]
        L.dpo_set_words.argtypes = [vp, u64p]
        L.dpo_set_count_intersection.restype = C.c_uint64
        L.dpo_set_count_intersection.argtypes = [vp, vp]
        L.dpo_set_count_intersection_to.restype = C.c_int64
        L.dpo_set_count_intersection_to.argtypes = [vp, vp, C.c_int64]
        L.dpo_shared_ids.restype = C.c_int64
        L.dpo_shared_ids.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int, u64p, C.c_int64]
        L.dpo_soft_union.argtypes = [C.c_int, u64p, C.c_int64, u64p]
        L.dpo_gap_range.argtypes = [C.c_int64, C.c_int, i64p]
        L.dpo_pairwise.argtypes = [i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_int, C.c_int64, i64p, i64p, i64p,
                                   C.c_int64, i64p]
        L.dpo_match.argtypes = [i64p, C.c_int64, i64p, C.c_int64, C.c_int64, C.c_int, i64p, i64p, i64p, C.c_int64, i64p]
        L.dpo_reads_from_fasta.restype = vp
        L.dpo_reads_from_fasta.argtypes = [C.c_char_p, C.c_int64, C.c_int]
        L.dpo_reads_from_arrays.restype = vp
        L.dpo_reads_from_arrays.argtypes = [C.c_char_p, i64p, C.c_int64, C.c_int64, C.c_int]
        L.dpo_reads_from_arrays_q.restype = vp
        L.dpo_reads_from_arrays_q.argtypes = [C.c_char_p, C.c_char_p, i64p, C.c_int64, C.c_int64, C.c_int]
        L.dpo_reads_free.argtypes = [vp]
        L.dpo_reads_count.restype = C.c_int64
        L.dpo_reads_count.argtypes = [vp]
        L.dpo_reads_reset_ignore.argtypes = [vp]
        L.dpo_reads_get_ignore.argtypes = [vp, u8p]
        L.dpo_kmer_values.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
        L.dpo_kmer_counts.argtypes = [vp, C.c_int, u64p]
        L.dpo_overlap_run.restype = vp
        L.dpo_overlap_run.argtypes = [vp, i64p, C.c_double, C.POINTER(C.c_double), C.c_int64, C.c_int]
        L.dpo_overlap_free.argtypes = [vp]
        L.dpo_overlap_rounds.restype = C.c_int64
        L.dpo_overlap_rounds.argtypes = [vp]
        for f in (L.dpo_overlap_paf, L.dpo_overlap_err, L.dpo_map_paf, L.dpo_map_err):
            f.restype = C.POINTER(C.c_char)
            f.argtypes = [vp, i64p]
        L.dpo_overlap_trace.restype = i64p
        L.dpo_overlap_trace.argtypes = [vp, C.c_int64, C.c_int, i64p]
        L.dpo_overlap_trace_paf.restype = C.POINTER(C.c_char)
        L.dpo_overlap_trace_paf.argtypes = [vp, C.c_int64, i64p]
        L.dpo_map_run.restype = vp
        L.dpo_map_run.argtypes = [vp, vp, i64p]
        L.dpo_map_free.argtypes = [vp]
        _lib = L
    return _lib


def synth():
    global _synth
    if _synth is None:
        if not os.path.exists(_SYNTH_SO):
            build_oracle()
        S = C.CDLL(_SYNTH_SO)
        S.dps_genome.argtypes = [C.c_uint64, C.c_int64, C.c_char_p]
        S.dps_reads.restype = C.c_int64
        S.dps_reads.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_int64,
                                i64p, i64p, u8p]
        _synth = S
    return _synth


def ptr(a, t):
    return a.ctypes.data_as(t)


def gen_reads(seed, G, N, L, e=0.0, variable=False):
    """Returns (bases uint8 array (ASCII), offsets int64[N+1])."""
    from tools.synth import gen_reads as _g
    b, off = _g(seed, G, N, L, e, variable)
    return b.copy(), off


def gen_genome(seed, G):
    from tools.synth import gen_genome as _g
    return _g(seed, G)


def write_fastq(path, bases, off, quals, prefix="r"):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            f.write(b"@%s%07d\n" % (prefix.encode(), i))
            f.write(bases[off[i]:off[i + 1]].tobytes())
            f.write(b"\n+\n")
            f.write(quals[off[i]:off[i + 1]].tobytes())
            f.write(b"\n")


def write_fasta(path, bases, off, prefix="r"):
    with open(path, "wb") as f:
        for i in range(len(off) - 1):
            f.write(b">%s%07d\n" % (prefix.encode(), i))
            f.write(bases[off[i]:off[i + 1]].tobytes())
            f.write(b"\n")


class Seq:
    def __init__(self, s=None, h=None):
        self.h = h if h is not None else lib().dpo_seq_new(s.encode(), len(s))

    def sub(self, a, b):
        return Seq(h=lib().dpo_seq_sub(self.h, a, b))

    def rc(self):
        return Seq(h=lib().dpo_seq_rc(self.h))

    def meta(self):
        m = np.zeros(6, dtype=np.int64)
        lib().dpo_seq_meta(self.h, ptr(m, i64p))
        return dict(nbytes=int(m[0]), firstLen=int(m[1]), finalLen=int(m[2]), offset=int(m[3]), inset=int(m[4]),
                    length=int(m[5]))

    def bytes(self):
        b = np.zeros(self.meta()["nbytes"], dtype=np.uint8)
        lib().dpo_seq_bytes(self.h, ptr(b, u8p))
        return b

    def __len__(self):
        return self.meta()["length"]

    def __str__(self):
        buf = C.create_string_buffer(self.meta()["nbytes"] * 4 + 8)
        n = lib().dpo_seq_str(self.h, buf, len(buf))
        return buf.raw[:n].decode()

    def kmer_at(self, i, k):
        return lib().dpo_seq_kmer_at(self.h, i, k)

    def next_kmer(self, cur, mask, idx):
        return lib().dpo_seq_next_kmer(self.h, cur, mask, idx)

    def count_kmers(self, upto, k, seeds):
        return lib().dpo_seq_count_kmers(self.h, upto, k, ptr(seeds, u8p))

    def count_kmers_between(self, a, b, upto, k, seeds):
        return lib().dpo_seq_count_kmers_between(self.h, a, b, upto, k, ptr(seeds, u8p))

    def write_segments(self, k, seeds):
        out = np.zeros(self.meta()["nbytes"] * 8 + 64, dtype=np.int64)
        n = lib().dpo_seq_write_segments(self.h, k, ptr(seeds, u8p), ptr(out, i64p))
        return out[:n].copy()


def byte_count_kmers(s, upto, k, seeds):
    return lib().dpo_byte_count_kmers(s.encode(), len(s), upto, k, ptr(seeds, u8p))


def byte_write_segments(s, k, seeds):
    out = np.zeros(len(s) * 2 + 8, dtype=np.int64)
    n = lib().dpo_byte_write_segments(s.encode(), len(s), k, ptr(seeds, u8p), ptr(out, i64p))
    return out[:n].copy()


def add_seeds_each(seqs, k, num_seeds, values):
    """AddSeeds of every Seq into an empty index: list of seedMaps (k-mers in seed-id order)."""
    L = lib()
    L.dpo_add_seeds_each.restype = C.c_int
    L.dpo_add_seeds_each.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p]
    n = len(seqs)
    hs = (C.c_void_p * n)(*[s.h for s in seqs])
    cap = n * num_seeds * 2 + 16
    out = np.zeros(cap, dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    vals = np.ascontiguousarray(values, dtype=np.float64)
    rc = L.dpo_add_seeds_each(hs, n, k, num_seeds, vals.ctypes.data, out.ctypes.data, cap, off.ctypes.data)
    if rc != 0:
        raise RuntimeError(L.dpo_last_error().decode())
    return [out[off[i]:off[i + 1]].copy() for i in range(n)]


def kmer_value(s):
    v = 0
    for ch in s.encode():
        v = (v << 2) | (((ch >> 1) ^ ((ch & 4) >> 2)) & 3)
    return v


class IntSet:
    def __init__(self, cap=None):
        self.h = lib().dpo_set_new() if cap is None else lib().dpo_set_new_cap(cap)

    def add(self, x):
        lib().dpo_set_add(self.h, x)

    def contains(self, x):
        return bool(lib().dpo_set_contains(self.h, x))

    def size(self):
        return lib().dpo_set_size(self.h)

    def window(self):
        w = np.zeros(3, dtype=np.uint64)
        lib().dpo_set_window(self.h, ptr(w, u64p))
        return int(w[0]), int(w[1]), int(w[2])

    def words(self):
        n = self.window()[2]
        w = np.zeros(n, dtype=np.uint64)
        lib().dpo_set_words(self.h, ptr(w, u64p))
        return w

    def count_intersection(self, o):
        return lib().dpo_set_count_intersection(self.h, o.h)

    def count_intersection_to(self, o, mx):
        return lib().dpo_set_count_intersection_to(self.h, o.h, mx)


def shared_ids(sets, min_count, fast):
    arr = (C.c_void_p * len(sets))(*[s.h for s in sets])
    out = np.zeros(1 << 16, dtype=np.uint64)
    n = lib().dpo_shared_ids(arr, len(sets), min_count, 1 if fast else 0, ptr(out, u64p), len(out))
    assert n >= 0
    return out[:n].copy()


def soft_union(which, words):
    w = np.ascontiguousarray(words, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().dpo_soft_union(which, ptr(w, u64p), len(w), ptr(out, u64p))
    return out


def _matches(fn, *args):
    cap = 1 << 16
    counts = np.zeros(1024, dtype=np.int64)
    a = np.zeros(cap, dtype=np.int64)
    b = np.zeros(cap, dtype=np.int64)
    n = C.c_int64(0)
    rc = fn(*args, ptr(counts, i64p), ptr(a, i64p), ptr(b, i64p), cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(lib().dpo_last_error().decode())
    out = []
    pos = 0
    for i in range(n.value):
        c = int(counts[i])
        out.append((a[pos:pos + c].copy(), b[pos:pos + c].copy()))
        pos += c
    return out


def pairwise(a_seg, b_seg, min_matches, k, max_length=500):
    a = np.ascontiguousarray(a_seg, dtype=np.int64)
    b = np.ascontiguousarray(b_seg, dtype=np.int64)
    return _matches(lib().dpo_pairwise, ptr(a, i64p), len(a), ptr(b, i64p), len(b), min_matches, k, max_length)


def match(seq_seg, q_seg, min_match, k):
    s = np.ascontiguousarray(seq_seg, dtype=np.int64)
    q = np.ascontiguousarray(q_seg, dtype=np.int64)
    return _matches(lib().dpo_match, ptr(s, i64p), len(s), ptr(q, i64p), len(q), min_match, k)


class ReadSet:
    def __init__(self, bases=None, off=None, min_len=0, himem=True, fasta=None, quals=None):
        """quals: raw FASTQ quality characters (uint8, same offsets as bases): the set then behaves like a FASTQ file's."""
        if fasta is not None:
            self.h = lib().dpo_reads_from_fasta(fasta.encode(), min_len, 1 if himem else 0)
            if not self.h:
                raise RuntimeError(lib().dpo_last_error().decode())
        else:
            b = np.ascontiguousarray(bases, dtype=np.uint8)
            o = np.ascontiguousarray(off, dtype=np.int64)
            if quals is not None:
                q = np.ascontiguousarray(quals, dtype=np.uint8)
                self.h = lib().dpo_reads_from_arrays_q(b.ctypes.data_as(C.c_char_p), q.ctypes.data_as(C.c_char_p), ptr(o, i64p),
                                                       len(o) - 1, min_len, 1 if himem else 0)
            else:
                self.h = lib().dpo_reads_from_arrays(b.ctypes.data_as(C.c_char_p), ptr(o, i64p), len(o) - 1, min_len,
                                                     1 if himem else 0)

    def __len__(self):
        return lib().dpo_reads_count(self.h)

    def reset_ignore(self):
        lib().dpo_reads_reset_ignore(self.h)

    def ignore(self):
        out = np.zeros(len(self), dtype=np.uint8)
        lib().dpo_reads_get_ignore(self.h, ptr(out, u8p))
        return out

    def kmer_values(self, k):
        v = np.zeros(4 ** k, dtype=np.float64)
        assert lib().dpo_kmer_values(self.h, k, v.ctypes.data_as(C.POINTER(C.c_double))) == 0
        return v

    def kmer_counts(self, k):
        v = np.zeros(4 ** k, dtype=np.uint64)
        assert lib().dpo_kmer_counts(self.h, k, ptr(v, u64p)) == 0
        return v


def _bytes(fn, h, *a):
    n = C.c_int64(0)
    p = fn(h, *a, C.byref(n))
    return C.string_at(p, n.value)


class OverlapRun:
    FIELDS = dict(seedKmers=0, queryIDs=1, querySeqIDs=2, indexedIds=3, indexedLength=4, indexedOffset=5,
                  indexedInset=6, matchQueryIndex=7, matchTarget=8, newlyIgnored=9, scalars=10, queryLength=11, queryOffset=12, queryInset=13, querySegments=20,
                  indexedSegments=21, candidates=22, matchA=23, matchB=24)

    def __init__(self, reads, k=10, overlap_size=1000, num_seeds=15, seed_batch_size=10000, chunk_size=10000,
                 query_batch_size=20000, min_hits=0.25, himem=True, values=None, max_rounds=-1, traces=False, query_type=1):
        p = np.array([overlap_size, k, num_seeds, seed_batch_size, chunk_size, query_batch_size,
                      (1 if himem else 0) | (query_type << 8)], dtype=np.int64)
        vp = values.ctypes.data_as(C.POINTER(C.c_double)) if values is not None else None
        self.h = lib().dpo_overlap_run(reads.h, ptr(p, i64p), float(min_hits), vp, max_rounds, 1 if traces else 0)
        if not self.h:
            raise RuntimeError(lib().dpo_last_error().decode())

    @property
    def rounds(self):
        return lib().dpo_overlap_rounds(self.h)

    @property
    def paf(self):
        return _bytes(lib().dpo_overlap_paf, self.h).decode()

    @property
    def err(self):
        return _bytes(lib().dpo_overlap_err, self.h).decode()

    def trace(self, rnd, name):
        f = self.FIELDS[name]
        n = C.c_int64(0)
        p = lib().dpo_overlap_trace(self.h, rnd, f, C.byref(n))
        assert n.value >= 0
        data = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)
        if f < 20:
            return data
        p = lib().dpo_overlap_trace(self.h, rnd, f + 100, C.byref(n))
        off = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        return data, off

    def trace_paf(self, rnd):
        return _bytes(lib().dpo_overlap_trace_paf, self.h, rnd).decode()


def map_run(ref, reads, circular=True, k=11, query_size=1000, min_length=500, chunk_size=10000, seed_rate=40):
    p = np.array([1 if circular else 0, k, query_size, min_length, chunk_size, seed_rate], dtype=np.int64)
    h = lib().dpo_map_run(ref.h, reads.h, ptr(p, i64p))
    if not h:
        raise RuntimeError(lib().dpo_last_error().decode())
    paf = _bytes(lib().dpo_map_paf, h).decode()
    err = _bytes(lib().dpo_map_err, h).decode()
    lib().dpo_map_free(h)
    return paf, err
