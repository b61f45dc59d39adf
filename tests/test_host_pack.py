"""`map` hands its reads to the device 2-bit packed, as the reference holds them (sequence/sequence.go:59-93 packBytes): the host library's
packer (AVX2 and scalar) against the oracle's NewPackedSequence (packBytesAsm for the whole groups + the Go tail), every length mod 128 and mod 4, any byte."""
import ctypes as C

import numpy as np

from tests import oracle_lib as O


def _pack(H, b, scalar):
    out = np.full((len(b) + 3) // 4 + 3, 0xA5, dtype=np.uint8)
    H.dph_pack_bases(b.tobytes(), len(b), out.ctypes.data, scalar)
    assert (out[(len(b) + 3) // 4:] == 0xA5).all()   # nothing written behind the read's last byte
    return out[:(len(b) + 3) // 4]


def test_packer_equals_reference_packbytes():
    from downpore_amd.overlap import load_host
    H = load_host()
    H.dph_pack_bases.restype = None
    H.dph_pack_bases.argtypes = [C.c_char_p, C.c_int64, C.c_void_p, C.c_int]
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTacgtNnRYKM-", dtype=np.uint8)
    for n in list(range(0, 140)) + [255, 256, 257, 511, 1000, 4099, 65537]:
        for any_byte in (False, True):
            b = np.ascontiguousarray(rng.integers(0, 256, n, dtype=np.uint8) if any_byte else alphabet[rng.integers(0, len(alphabet), n)])
            want = O.Seq(h=O.lib().dpo_seq_new(b.tobytes(), n)).bytes() if n else np.zeros(0, dtype=np.uint8)   # NewPackedSequence
            assert len(want) == (n + 3) // 4
            if n % 4:   # the last byte's unused bits are zero
                assert want[-1] & ((1 << (2 * (4 - n % 4))) - 1) == 0
            for scalar in (0, 1):
                assert np.array_equal(_pack(H, b, scalar), want), (n, any_byte, scalar)
