"""CPU test of the product's Planner (libdownpore_host.so, host seed selection, no GPU): ignore flags that arrive while
the planner thread holds a finished plan must invalidate it when ANY flagged read id is >= the plan's firstIn, also when
the same commit flags a smaller id as well (the round-1 code compared the smallest flagged id)."""
import ctypes as C
import os

import numpy as np

from tests import oracle_lib as O


def test_plan_computed_before_flags_is_discarded():
    os.environ["DP_TUNE"] = "plan_delay_us=300000"
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
        del os.environ["DP_TUNE"]


def test_touch_test_vector_variants_agree():
    """SeedIndex::touchesSeed(kmers, n) - the planner's probe of a window's evaluated k-mers against the seeds committed so far -
    has a scalar, an AVX2 and an AVX-512 form (pre-filter probes gathered 8 / 16 at a time): all must give the scalar answer, for
    windows with no seed, with one seed in any lane (also in the scalar tail), and with unused 0xffffffff slots."""
    from downpore_amd.overlap import load_host
    H = load_host()
    k, stride, n_windows = 13, 493, 600  # stride not a multiple of 16: the tail loop runs
    rng = np.random.default_rng(5)
    seeds = np.unique(rng.integers(1, 4 ** k, 20000)).astype(np.uint32)
    kmers = rng.integers(1, 4 ** k, (n_windows, stride)).astype(np.uint32)
    kmers[:, 470:] = 0xffffffff  # unused slots at the end of every window
    want = np.zeros(n_windows, dtype=np.uint8)
    seedset = set(seeds.tolist())
    for w in range(0, n_windows, 3):  # plant one seed at a lane that walks through every position, incl. the tail
        kmers[w, (w * 7) % 470] = seeds[(w * 31) % len(seeds)]
    for w in range(n_windows):
        want[w] = 1 if any(int(x) in seedset for x in kmers[w, :470]) else 0
    kmers = np.ascontiguousarray(kmers)
    res = np.zeros(n_windows, dtype=np.uint8)
    mask = C.c_int(0)
    H.dph_selftest_touch.restype = C.c_int
    H.dph_selftest_touch.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int)]
    bad = H.dph_selftest_touch(k, seeds.ctypes.data, len(seeds), kmers.ctypes.data, n_windows, stride, res.ctypes.data, C.byref(mask))
    assert bad == 0
    assert np.array_equal(res, want)
    assert want.sum() >= n_windows // 3


def test_planner_lanes_compute_the_chain_of_one_lane():
    """Planner lanes start plans from guesses of where their predecessors end.  The whole plan chain of a read set
    - windows, seed lists, firstSequence of every round - must come out the same from the general PrepareQueries path, from the
    window-cache path with one lane and from the window-cache path with several lanes, with reads flagged along the way.  The
    window cache's producer selects on the host here (no GPU).  k = 8 with a large seed budget: a sixth of all k-mers are seeds,
    nearly every window is re-selected and reverse complements collide, so guesses fail and plans are recomputed."""
    from downpore_amd.overlap import Reads, load_host
    H = load_host()
    H.dph_selftest_planner_lanes.restype = C.c_int
    H.dph_selftest_planner_lanes.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
    H.dph_planner_counter.restype = C.c_int64
    H.dph_planner_counter.argtypes = [C.c_int]
    for (seed, G, N, L, variable, k, budget, lanes, flag_every) in [(7, 60000, 400, 3000, False, 8, 1500, 4, 0),
                                                                    (8, 80000, 500, 2500, True, 10, 800, 3, 3),
                                                                    (9, 50000, 300, 4000, False, 8, 4000, 5, 2)]:
        bases, off = O.gen_reads(seed, G, N, L, 0.0, variable)
        reads = Reads(bases, off, min_len=1000)
        rng = np.random.default_rng(seed)
        values = np.ascontiguousarray(rng.random(4 ** k))
        values[0] = 0.0
        n = C.c_int64(0)
        thrown = H.dph_planner_counter(1)
        rc = H.dph_selftest_planner_lanes(reads.h, k, budget, values.ctypes.data, lanes, flag_every, C.byref(n))
        assert rc == 0, "chains differ (rc %d; > 0: first differing round + 1) for k=%d budget=%d lanes=%d" % (rc, k, budget, lanes)
        assert n.value >= 5, "only %d plans: the case does not exercise the chain" % n.value
        print("k=%d budget=%d lanes=%d: %d plans, %d computed plans thrown away" % (k, budget, lanes, n.value, H.dph_planner_counter(1) - thrown))


def test_planner_ownership_computes_a_rank_share_of_the_chain():
    """Round-parallel runs: every rank's planner computes the plans of the rounds its rank executes only and guesses
    where the rounds in between end (Planner::setOwnership).  Simulated ranks - each with its own flags, window cache (host
    producer) and planner - play a commit that accepts a plan iff it starts at the committed firstSequence and otherwise tells
    every rank the truth and asks the owner again.  The accepted chain must be the chain one dense planner walks: with guesses
    that hold (k = 10, small budget), with guesses that mostly fail (k = 8: a sixth of all k-mers are seeds) and with reads
    flagged along the way."""
    from downpore_amd.overlap import Reads, load_host
    H = load_host()
    H.dph_selftest_planner_sparse.restype = C.c_int
    H.dph_selftest_planner_sparse.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    for (seed, G, N, L, variable, k, budget, world, flag_every) in [(8, 80000, 500, 2500, True, 10, 800, 2, 0), (8, 80000, 500, 2500, True, 10, 800, 4, 3),
                                                                    (7, 60000, 400, 3000, False, 8, 1500, 3, 0), (9, 50000, 300, 4000, False, 8, 4000, 8, 2)]:
        bases, off = O.gen_reads(seed, G, N, L, 0.0, variable)
        reads = Reads(bases, off, min_len=1000)
        rng = np.random.default_rng(seed)
        values = np.ascontiguousarray(rng.random(4 ** k))
        values[0] = 0.0
        n, redone = C.c_int64(0), C.c_int64(0)
        rc = H.dph_selftest_planner_sparse(reads.h, k, budget, values.ctypes.data, world, flag_every, C.byref(n), C.byref(redone))
        assert rc == 0, "chains differ (rc %d; > 0: first differing round + 1) for k=%d budget=%d world=%d" % (rc, k, budget, world)
        assert n.value >= 5
        print("k=%d budget=%d world=%d flags every %d: %d plans, %d asked for again" % (k, budget, world, flag_every, n.value, redone.value))
