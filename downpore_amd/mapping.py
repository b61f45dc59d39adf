"""`downpore map` on the GPU: Python front end of the C++ host mirror of mapping.Mapper (libdownpore_host.so)."""
import ctypes as C

import numpy as np

from .hip import DpError
from .overlap import load_host

MAP_STAT_FIELDS = ["n_chunks", "n_seeds", "n_windows", "n_chains", "n_batches", "k_scan_ms", "k_map_ms", "t_setup_s", "t_scan_s",
                   "t_chain_s", "t_host_s", "map_bytes", "scan_bytes"]


def map_reads(ref, reads, circular=True, k=11, query_size=1000, min_length=500, chunk_size=10000, seed_rate=40, device=0):
    """ref / reads: downpore_amd.overlap.Reads (reference loaded with min_len=0, reads with min_len=min_length; both are
    treated as top-level sequences exactly like commands/map.go does).  Returns (paf, stderr_text, stats)."""
    H = load_host()
    H.dph_map_run.restype = C.c_void_p
    H.dph_map_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    H.dph_map_free.argtypes = [C.c_void_p]
    for f in (H.dph_map_paf, H.dph_map_errtext):
        f.restype = C.POINTER(C.c_char)
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    H.dph_map_stats.argtypes = [C.c_void_p, C.c_void_p]
    p = np.array([1 if circular else 0, k, query_size, min_length, chunk_size, seed_rate], dtype=np.int64)
    h = H.dph_map_run(ref.h, reads.h, p.ctypes.data, device)
    if not h:
        raise DpError("dph_map_run: " + H.dph_last_error(None).decode())
    n = C.c_int64(0)
    paf = C.string_at(H.dph_map_paf(h, C.byref(n)), n.value).decode()
    err = C.string_at(H.dph_map_errtext(h, C.byref(n)), n.value).decode()
    st = np.zeros(len(MAP_STAT_FIELDS), dtype=np.float64)
    H.dph_map_stats(h, st.ctypes.data)
    H.dph_map_free(h)
    return paf, err, dict(zip(MAP_STAT_FIELDS, st.tolist()))
