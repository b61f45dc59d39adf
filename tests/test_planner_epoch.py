"""CPU test of the product's Planner (libdownpore_host.so, host seed selection, no GPU): ignore flags that arrive while
the planner thread holds a finished plan must invalidate it when ANY flagged read id is >= the plan's firstIn, also when
the same commit flags a smaller id as well (the round-1 code compared the smallest flagged id)."""
import ctypes as C
import os

import numpy as np

from tests import oracle_lib as O


def test_plan_computed_before_flags_is_discarded():
    os.environ["DPH_TEST_PLAN_DELAY_US"] = "300000"
    try:
        from downpore_amd.overlap import Reads, load_host
        H = load_host()
        bases, off = O.gen_reads(7, 40000, 200, 3000, 0.0, False)
        reads = Reads(bases, off, min_len=1000)
        k = 8
        rng = np.random.default_rng(3)
        values = np.ascontiguousarray(rng.random(4 ** k))
        values[0] = 0.0
        H.dph_selftest_planner_flags.restype = C.c_int
        H.dph_selftest_planner_flags.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
        rc = H.dph_selftest_planner_flags(reads.h, k, 600, values.ctypes.data)
        assert rc == 0, "planner handed out a plan computed before the flags were set (rc %d)" % rc
    finally:
        del os.environ["DPH_TEST_PLAN_DELAY_US"]
