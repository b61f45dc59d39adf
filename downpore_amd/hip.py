"""ctypes binding of libdownpore_hip.so (C ABI: include/downpore_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_dir():
    # DP_LIB_DIR: another build of both libraries (make LIB=../lib_prof PROF=1, make LIB=../lib_copylog COPYLOG=1: diagnosis builds)
    return os.environ.get("DP_LIB_DIR") or os.path.join(_HERE, "lib")


def lib_path():
    return os.path.join(lib_dir(), "libdownpore_hip.so")


class DpError(RuntimeError):
    pass


class SeedSeqBatch(C.Structure):
    _fields_ = [("n_items", C.c_uint32), ("n_seeds", C.POINTER(C.c_uint32)), ("seg_off", C.POINTER(C.c_uint64)),
                ("segs", C.POINTER(C.c_int32)), ("n_segs", C.c_uint64), ("kernel_ms", C.c_double),
                ("count_kernel_ms", C.c_double), ("write_kernel_ms", C.c_double), ("bases_scanned", C.c_uint64)]


class SurvivorBatch(C.Structure):
    _fields_ = [("n_survivors", C.c_uint32), ("read", C.POINTER(C.c_uint32)), ("n_seeds", C.POINTER(C.c_uint32)),
                ("seg_off", C.POINTER(C.c_uint64)), ("n_extra", C.c_uint32), ("extra_n_seeds", C.POINTER(C.c_uint32)),
                ("extra_seg_off", C.POINTER(C.c_uint64)), ("segs", C.POINTER(C.c_int32)), ("n_segs", C.c_uint64),
                ("kernel_ms", C.c_double), ("count_kernel_ms", C.c_double), ("write_kernel_ms", C.c_double),
                ("bases_scanned", C.c_uint64), ("reads_scanned", C.c_uint32), ("index_mode", C.c_uint32),
                ("index_hits", C.c_uint64)]


class MatchBatch(C.Structure):
    _fields_ = [("n_matches", C.c_uint32), ("query", C.POINTER(C.c_uint32)), ("target", C.POINTER(C.c_uint32)),
                ("off", C.POINTER(C.c_uint64)), ("match_a", C.POINTER(C.c_int32)), ("match_b", C.POINTER(C.c_int32)),
                ("target_anchor", C.POINTER(C.c_int32)), ("n_queries", C.c_uint32), ("cand_off", C.POINTER(C.c_uint64)), ("cand", C.POINTER(C.c_uint32)),
                ("query_kernel_ms", C.c_double), ("chain_kernel_ms", C.c_double), ("query_bytes", C.c_uint64),
                ("chain_bytes", C.c_uint64)]


class ChainBatch(C.Structure):
    _fields_ = [("n_chains", C.c_uint32), ("window", C.POINTER(C.c_uint32)), ("target", C.POINTER(C.c_uint32)),
                ("off", C.POINTER(C.c_uint64)), ("match_a", C.POINTER(C.c_int32)), ("match_b", C.POINTER(C.c_int32)),
                ("kernel_ms", C.c_double)]


#: every entry point include/downpore_hip.h declares (checked by the CPU-side symbol test)
SYMBOLS = ["dp_version", "dp_ctx_create", "dp_ctx_create_shared", "dp_ctx_set_priority", "dp_ctx_destroy", "dp_last_error", "dp_reads_upload", "dp_reads_packed",
           "dp_reads_count", "dp_reads_total_bases", "dp_kmer_histogram", "dp_kmer_values", "dp_round_begin", "dp_scan", "dp_scan_prepare", "dp_scan_reads", "dp_index_build",
           "dp_find_overlaps", "dp_query_prestage", "dp_map_windows", "dp_index_posting_row", "dp_index_seedset_row", "dp_scan_device_buffers",
           "dp_scan_import_segments", "dp_values_upload", "dp_select_seeds", "dp_reads_upload_rc", "dp_consensus_align", "dp_scan_release", "dp_consensus_paf", "dp_fetch_overlaps", "dp_select_windows", "dp_values_download",
    "dp_values_download_codes", "dp_values_download_codes8", "dp_index_build_chunked", "dp_index_prechain", "dp_index_prechained", "dp_index_chunks", "dp_scan_fetch_mode", "dp_scan_fetch_segments", "dp_set_stream_wait", "dp_set_kernel_timing", "dp_index_meta", "dp_index_set_global", "dp_map_windows_shard", "dp_single_seed_candidates", "dp_comm_unique_id", "dp_comm_init", "dp_comm_init_local", "dp_quality_upload",
           "dp_comm_destroy", "dp_comm_abort", "dp_allgather_blobs", "dp_gather_blobs", "dp_kindex_set_comm", "dp_kindex_digest", "dp_release_device_caches", "dp_reads_upload_rc_begin", "dp_reads_upload_wait", "dp_reads_upload_packed_rc", "dp_host_alloc", "dp_host_free", "dp_comm_rank", "dp_comm_size", "dp_allgather_survivors"]

_lib = None


def load_library():
    """Loads the HIP library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise DpError("libdownpore_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C downpore_amd/csrc`. There is no CPU fallback." % p)
    L = C.CDLL(p)
    vp = C.c_void_p
    L.dp_version.restype = C.c_char_p
    L.dp_last_error.restype = C.c_char_p
    L.dp_last_error.argtypes = [vp]
    L.dp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.dp_ctx_create_shared.argtypes = [vp, C.POINTER(vp)]
    L.dp_ctx_destroy.argtypes = [vp]
    L.dp_reads_upload.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_uint32]
    L.dp_reads_packed.argtypes = [vp, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.dp_reads_count.restype = C.c_uint32
    L.dp_reads_count.argtypes = [vp]
    L.dp_reads_total_bases.restype = C.c_uint64
    L.dp_reads_total_bases.argtypes = [vp]
    L.dp_kmer_histogram.argtypes = [vp, C.c_int, C.c_void_p]
    L.dp_kmer_values.argtypes = [vp, C.c_int, C.c_void_p]
    L.dp_round_begin.argtypes = [vp, C.c_int, C.c_void_p, C.c_uint32]
    L.dp_scan.argtypes = [vp, C.c_void_p, C.c_uint32, C.POINTER(SeedSeqBatch)]
    L.dp_scan_reads.argtypes = [vp, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_uint32,
                                C.POINTER(SurvivorBatch)]
    L.dp_index_build.argtypes = [vp, C.c_void_p, C.c_uint32]
    L.dp_find_overlaps.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_uint32, C.c_double, C.c_int, C.c_uint32, C.c_int,
                                   C.POINTER(MatchBatch)]
    L.dp_map_windows.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(ChainBatch)]
    L.dp_index_posting_row.argtypes = [vp, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.dp_index_seedset_row.argtypes = [vp, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.dp_scan_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.dp_scan_import_segments.argtypes = [vp, C.c_void_p, C.c_uint64]
    _lib = L
    return L


def _arr(p, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(p, shape=(int(n),)).astype(dtype, copy=True)


class Context:
    """One GPU context (`dp_ctx`): reads resident in HBM + per-round seed/index state."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.dp_ctx_create(device, C.byref(h))
        if rc != 0:
            raise DpError("dp_ctx_create failed (%d): %s" % (rc, self.L.dp_last_error(None).decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.dp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise DpError("libdownpore_hip error %d: %s" % (rc, self.L.dp_last_error(self.h).decode()))

    # ---- A1
    def upload_reads(self, bases, off):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        self._chk(self.L.dp_reads_upload(self.h, bases.ctypes.data, off.ctypes.data, len(off) - 1))
        self.read_len = np.diff(off).astype(np.int64)

    def upload_reads_rc(self, bases, off, first_paired):
        """Reads >= first_paired are stored as (forward, reverse complement) pairs; the device makes the second strand."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        self.L.dp_reads_upload_rc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        self._chk(self.L.dp_reads_upload_rc(self.h, bases.ctypes.data, off.ctypes.data, len(off) - 1, first_paired))
        ln = np.diff(off).astype(np.int64)
        self.read_len = np.concatenate([ln[:first_paired], np.repeat(ln[first_paired:], 2)])

    def upload_reads_packed_rc(self, packed, lens, first_paired, pinned=False):
        """The reads arrive 2-bit packed (read r at the sum of the 16-byte-rounded packed sizes before it); pinned: through a block of
        dp_host_alloc, as the mapper does."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.L.dp_reads_upload_packed_rc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        self.L.dp_host_alloc.restype = C.c_void_p
        self.L.dp_host_alloc.argtypes = [C.c_size_t]
        self.L.dp_host_free.argtypes = [C.c_void_p]
        blk = None
        ptr = packed.ctypes.data
        if pinned:
            blk = self.L.dp_host_alloc(max(1, packed.size))
            assert blk
            C.memmove(blk, packed.ctypes.data, packed.size)
            ptr = blk
        try:
            self._chk(self.L.dp_reads_upload_packed_rc(self.h, ptr, lens.ctypes.data, len(lens), first_paired))
        finally:
            if blk:
                self.L.dp_host_free(blk)
        ln = lens.astype(np.int64)
        self.read_len = np.concatenate([ln[:first_paired], np.repeat(ln[first_paired:], 2)])

    def packed_read(self, r):
        nb = (int(self.read_len[r]) + 3) // 4
        out = np.zeros(max(nb, 1), dtype=np.uint8)
        n = C.c_uint64(0)
        self._chk(self.L.dp_reads_packed(self.h, r, out.ctypes.data, nb, C.byref(n)))
        return out[:nb]

    # ---- A22
    def kmer_histogram(self, k):
        out = np.zeros(4 ** k, dtype=np.uint64)
        self._chk(self.L.dp_kmer_histogram(self.h, k, out.ctypes.data))
        return out

    # ---- A22 + A23: value table computed (and left resident) on the device
    def kmer_values(self, k):
        out = np.zeros(4 ** k, dtype=np.float64)
        self._chk(self.L.dp_kmer_values(self.h, k, out.ctypes.data))
        return out

    # ---- round
    def round_begin(self, k, seed_kmers):
        s = np.ascontiguousarray(seed_kmers, dtype=np.uint32)
        self.k = k
        self.n_seeds = len(s)
        self._chk(self.L.dp_round_begin(self.h, k, s.ctypes.data, len(s)))

    # ---- A2 + A10
    def scan(self, items):
        """items: array-like of (read, start, n_kmers, min_seeds).  Returns dict with n_seeds, seg_off, segs."""
        it = np.ascontiguousarray(items, dtype=np.uint32).reshape(-1, 4)
        b = SeedSeqBatch()
        self._chk(self.L.dp_scan(self.h, it.ctypes.data, len(it), C.byref(b)))
        n = b.n_items
        return dict(n_seeds=_arr(b.n_seeds, n, np.uint32), seg_off=_arr(b.seg_off, n + 1, np.uint64),
                    segs=_arr(b.segs, b.n_segs, np.int32), kernel_ms=b.kernel_ms, count_kernel_ms=b.count_kernel_ms,
                    write_kernel_ms=b.write_kernel_ms, bases_scanned=b.bases_scanned)

    def scan_prepare(self, k):
        """One-off work of dp_scan_reads (the resident k-mer position index, when this read set gets one) done now."""
        self.L.dp_scan_prepare.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.L.dp_scan_prepare(self.h, k))

    def set_priority(self, high=True):
        self.L.dp_ctx_set_priority.argtypes = [C.c_void_p, C.c_int]
        self._chk(self.L.dp_ctx_set_priority(self.h, 1 if high else 0))

    def scan_reads(self, ignore, epoch, lo, hi, top_level, min_seeds, extra=None):
        ig = np.ascontiguousarray(ignore, dtype=np.uint8)
        ex = np.ascontiguousarray(extra if extra is not None else np.zeros((0, 4)), dtype=np.uint32).reshape(-1, 4)
        b = SurvivorBatch()
        self._chk(self.L.dp_scan_reads(self.h, ig.ctypes.data, epoch, lo, hi, 1 if top_level else 0, min_seeds, ex.ctypes.data,
                                       len(ex), C.byref(b)))
        ns, ne = b.n_survivors, b.n_extra
        return dict(read=_arr(b.read, ns, np.uint32), n_seeds=_arr(b.n_seeds, ns, np.uint32), seg_off=_arr(b.seg_off, ns, np.uint64),
                    extra_n_seeds=_arr(b.extra_n_seeds, ne, np.uint32), extra_seg_off=_arr(b.extra_seg_off, ne, np.uint64),
                    segs=_arr(b.segs, b.n_segs, np.int32), bases_scanned=b.bases_scanned, reads_scanned=b.reads_scanned,
                    kernel_ms=b.kernel_ms, index_mode=int(b.index_mode), index_hits=int(b.index_hits))

    # ---- A9 selection
    def values_upload(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self.L.dp_values_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        self._chk(self.L.dp_values_upload(self.h, v.ctypes.data, len(v)))

    def select_seeds(self, windows, k, num_seeds):
        """windows: (read, start, length in bases) rows -> uint32 [n, num_seeds] k-mers in insertion-list order."""
        w = np.zeros((len(windows), 4), dtype=np.uint32)
        if len(windows):
            w[:, :3] = np.asarray(windows, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros((len(windows), num_seeds), dtype=np.uint32)
        self.L.dp_select_seeds.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        self._chk(self.L.dp_select_seeds(self.h, w.ctypes.data, len(w), k, num_seeds, out.ctypes.data))
        return out

    def import_segments(self, segs):
        s = np.ascontiguousarray(segs, dtype=np.int32)
        self._chk(self.L.dp_scan_import_segments(self.h, s.ctypes.data, len(s)))

    def scan_device_buffer(self):
        p = C.c_void_p()
        n = C.c_uint64(0)
        self._chk(self.L.dp_scan_device_buffers(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- A13
    def index_build(self, seg_off, n_seeds):
        refs = np.zeros(len(seg_off), dtype=[("seg_off", np.uint64), ("n_seeds", np.uint32), ("reserved", np.uint32)])
        refs["seg_off"] = seg_off
        refs["n_seeds"] = n_seeds
        self._chk(self.L.dp_index_build(self.h, refs.ctypes.data, len(refs)))
        self.n_seqs = len(refs)

    def posting_row(self, seed):
        W = max(1, (self.n_seqs + 63) // 64)
        words = np.zeros(W, dtype=np.uint64)
        nw, cnt, st, en = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.dp_index_posting_row(self.h, seed, words.ctypes.data, W, C.byref(nw), C.byref(cnt), C.byref(st),
                                              C.byref(en)))
        return words, cnt.value, st.value, en.value

    def seedset_row(self, seq):
        SW = max(1, (self.n_seeds + 63) // 64)
        words = np.zeros(SW, dtype=np.uint64)
        nw = C.c_uint32()
        self._chk(self.L.dp_index_seedset_row(self.h, seq, words.ctypes.data, SW, C.byref(nw)))
        return words

    # ---- A14 .. A8
    def find_overlaps(self, q_segs, q_off, hit_fraction, k, max_query_len, want_candidates=False):
        qs = np.ascontiguousarray(q_segs, dtype=np.int32)
        qo = np.ascontiguousarray(q_off, dtype=np.uint64)
        b = MatchBatch()
        self._chk(self.L.dp_find_overlaps(self.h, qs.ctypes.data, qo.ctypes.data, len(qo) - 1, float(hit_fraction), k,
                                          max_query_len, 1 if want_candidates else 0, C.byref(b)))
        nm = b.n_matches
        off = _arr(b.off, nm + 1, np.uint64)
        tot = int(off[-1]) if nm else 0
        res = dict(query=_arr(b.query, nm, np.uint32), target=_arr(b.target, nm, np.uint32), off=off,
                   match_a=_arr(b.match_a, tot, np.int32), match_b=_arr(b.match_b, tot, np.int32),
                   target_anchor=_arr(b.target_anchor, 2 * nm, np.int32).reshape(-1, 2),
                   query_kernel_ms=b.query_kernel_ms, chain_kernel_ms=b.chain_kernel_ms, query_bytes=b.query_bytes,
                   chain_bytes=b.chain_bytes)
        if want_candidates:
            co = _arr(b.cand_off, b.n_queries + 1, np.uint64)
            res["cand_off"] = co
            res["cand"] = _arr(b.cand, int(co[-1]), np.uint32)
        return res

    # ---- A19 + A20
    def map_windows(self, w_segs, w_off, w_len, k):
        ws = np.ascontiguousarray(w_segs, dtype=np.int32)
        wo = np.ascontiguousarray(w_off, dtype=np.uint64)
        wl = np.ascontiguousarray(w_len, dtype=np.uint32)
        b = ChainBatch()
        self._chk(self.L.dp_map_windows(self.h, ws.ctypes.data, wo.ctypes.data, wl.ctypes.data, len(wo) - 1, k, C.byref(b)))
        n = b.n_chains
        off = _arr(b.off, n + 1, np.uint64)
        tot = int(off[-1]) if n else 0
        return dict(window=_arr(b.window, n, np.uint32), target=_arr(b.target, n, np.uint32), off=off,
                    match_a=_arr(b.match_a, tot, np.int32), match_b=_arr(b.match_b, tot, np.int32), kernel_ms=b.kernel_ms)
