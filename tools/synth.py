"""ctypes wrapper of the seeded synthetic-read generator (tools/synth.cpp) — neutral tooling shared by tests and bench."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdpsynth.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "synth.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", _SO])
        S = C.CDLL(_SO)
        S.dps_genome.argtypes = [C.c_uint64, C.c_int64, C.c_char_p]
        S.dps_reads.restype = C.c_int64
        S.dps_reads.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_void_p, C.c_int64,
                                C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = S
    return _lib


def gen_reads(seed, G, N, L, e=0.0, variable=False):
    """Returns (bases: uint8 ASCII array, off: int64[N+1])."""
    cap = int(N * (L * (2 if (variable or e > 0) else 1) + 16))
    bases = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(N + 1, dtype=np.int64)
    n = _load().dps_reads(seed, G, N, L, float(e), 1 if variable else 0, bases.ctypes.data, cap, off.ctypes.data, None, None)
    if n < 0:
        raise RuntimeError("synthetic read buffer too small")
    return bases[:n], off


def gen_genome(seed, G):
    buf = C.create_string_buffer(G)
    _load().dps_genome(seed, G, buf)
    return buf.raw


def gen_reads_truth(seed, G, N, L, e=0.0, variable=False):
    """gen_reads plus where every read came from: (bases, off, starts int64[N] = genome position of the read's template, strands
    uint8[N] = 1 when the read is the reverse complement of the genome).  The reads are the same as gen_reads'."""
    cap = int(N * (L * (2 if (variable or e > 0) else 1) + 16))
    bases = np.zeros(cap, dtype=np.uint8)
    off = np.zeros(N + 1, dtype=np.int64)
    starts = np.zeros(N, dtype=np.int64)
    strands = np.zeros(N, dtype=np.uint8)
    n = _load().dps_reads(seed, G, N, L, float(e), 1 if variable else 0, bases.ctypes.data, cap, off.ctypes.data, starts.ctypes.data,
                          strands.ctypes.data)
    if n < 0:
        raise RuntimeError("synthetic read buffer too small")
    return bases[:n], off, starts, strands
