"""End-to-end GPU parity: the product's `overlap` pipeline (host C++ above the C ABI + HIP kernels) must print the
same PAF as the ORACLE, round by round, and flag the same reads as ignored."""
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def first_diff(a, b):
    """None if equal; otherwise a short description (never hand two huge strings to pytest's differ)."""
    if a == b:
        return None
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got  %s\n  want %s" % (i, x, y)
    return "line counts differ: got %d want %d" % (len(la), len(lb))


def _run_both(seed, G, N, L, k, e=0.0, variable=False, himem=True, max_rounds=-1, slots=1, **kw):
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(seed, G, N, L, e, variable)
    rs = O.ReadSet(bases, off, min_len=kw.get("overlap_size", 1000), himem=himem)
    # a step of the pipeline may commit several rounds: let the oracle run a few rounds past max_rounds
    orun = O.OverlapRun(rs, k=k, himem=himem, max_rounds=max_rounds + 8 if max_rounds >= 0 else -1, traces=True, **kw)
    reads = Reads(bases, off, min_len=kw.get("overlap_size", 1000), himem=himem)
    pipe = OverlapPipeline(reads, k=k, himem=himem, slots=slots, **kw)
    rounds = 0
    while max_rounds < 0 or rounds < max_rounds:
        c = pipe.step()
        if c == 0:
            break
        assert rounds + c <= orun.rounds
        want = "".join(orun.trace_paf(r) for r in range(rounds, rounds + c))
        d = first_diff(pipe.round_paf(), want)
        assert d is None, "PAF differs in rounds %d..%d: %s" % (rounds, rounds + c - 1, d)
        rounds += c
    if max_rounds < 0:
        assert rounds == orun.rounds
        assert first_diff(pipe.all_paf(), orun.paf) is None
        assert np.array_equal(reads.ignore(), rs.ignore())
    else:
        assert first_diff(pipe.all_paf(), "".join(orun.trace_paf(r) for r in range(rounds))) is None
        flagged = sorted(set(int(x) for r in range(rounds) for x in orun.trace(r, "newlyIgnored")))
        assert sorted(np.nonzero(reads.ignore())[0].tolist()) == flagged
    st = pipe.stats()
    pipe.close()
    return orun, st


@pytest.mark.parametrize("k,G,N,L,e,variable", [(10, 100000, 400, 5000, 0.0, False), (10, 80000, 300, 6000, 0.03, True),
                                                 (13, 1500000, 3000, 10000, 0.0, False),
                                                 (13, 1200000, 2000, 12000, 0.002, True)])
def test_overlap_paf_bit_exact(k, G, N, L, e, variable):
    orun, st = _run_both(100 + k, G, N, L, k, e, variable, max_rounds=4)
    assert orun.paf.count("\n") > 0


@pytest.mark.parametrize("slots", [1, 4])
def test_overlap_full_run_config1_k10(slots):
    """BASELINE config 1 shape (1k reads x 5 kb) at the command's default k=10, all rounds; with 4 executor slots the
    rounds run concurrently (speculating on ignore flags) and are committed in order."""
    orun, st = _run_both(1, 250000, 1000, 5000, 10, slots=slots)
    assert orun.rounds >= 5


def test_config1_at_its_stated_k13():
    """BASELINE config 1 as BASELINE.json states it: 1 000 reads x 5 kb, k = 13.  At this size the value table's top-"2 %" cut
    (commands/overlap.go:73-93) removes every k-mer that occurs at all, so no window finds a seed: the reference's command runs one
    round and prints nothing (HISTORY.md 2, note 7).  The product must do exactly that - same round count, empty PAF, no read flagged -
    not merely at the command's default k = 10 where the other config-1 tests run."""
    orun, st = _run_both(1, 250000, 1000, 5000, 13, slots=1)
    assert orun.paf == "" and orun.rounds <= 1


@pytest.mark.parametrize("query_type,slots", [(4, 1), (4, 3), (2, 1), (9, 1), (12, 2)])
def test_overlap_other_query_types(query_type, slots):
    """PrepareQueries' other window layouts behind the same boundary (overlap.go:18-21,91-155): QueryAll=4 is what the
    `correct` command uses (commands/correct.go:97,169), QueryCentre=2, and WeightEdges=8 (seeds from the two 200-base
    sides of each window, numSeeds halved)."""
    orun, st = _run_both(58, 90000, 350, 4200, 10, e=0.01, variable=True, max_rounds=3, slots=slots, query_type=query_type)
    assert orun.rounds >= 1


def test_round_parallel_commits_beyond_local_plans():
    """A rank commits rounds that OTHER ranks executed: its own planner never computed their plans and must restart the
    chain from the committed firstSequence (4 ranks x 4 slots = 16 rounds per superstep, more than the planner's
    prefetch depth; this configuration used to hang)."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    world, slots = 4, 4
    bases, off = O.gen_reads(9, 500000, 2000, 5000, 0.0, False)
    kw = dict(k=10, seed_batch_size=1500)  # small seed budget: ~25 reads per round, many rounds
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], rank=r, world=world, mode="round-batch", slots=slots, **kw) for r in range(world)]
    committed = 0
    while committed < 48 and not pipes[0].finished():
        base = pipes[0].committed_rounds()
        blobs = [pipes[r].exec_round_blob(base + r * slots) for r in range(world)]
        cs = [pipes[r].commit_blobs(blobs) for r in range(world)]
        assert len(set(cs)) == 1 and cs[0] > 0
        committed += cs[0]
    rs = O.ReadSet(bases, off, min_len=1000)
    orun = O.OverlapRun(rs, max_rounds=committed, **kw)
    assert orun.rounds == committed
    for r in range(world):
        assert first_diff(pipes[r].all_paf(), orun.paf) is None
        assert pipes[r].committed_rounds() == committed
        pipes[r].close()


def _run_arrays(bases, off, k=10, slots=1, max_rounds=-1, **kw):
    from downpore_amd.overlap import OverlapPipeline, Reads
    rs = O.ReadSet(bases, off, min_len=1000)
    orun = O.OverlapRun(rs, k=k, max_rounds=max_rounds, **kw)
    reads = Reads(bases, off, min_len=1000)
    pipe = OverlapPipeline(reads, k=k, slots=slots, **kw)
    n = pipe.run(max_rounds)
    d = first_diff(pipe.all_paf(), orun.paf)
    assert d is None, d
    assert np.array_equal(reads.ignore(), rs.ignore())
    if max_rounds < 0:
        assert n == orun.rounds
    pipe.close()
    return orun


def test_overlap_degenerate_inputs():
    """Edge cases of the input set: nothing survives the length filter, a single read, two reads."""
    bases, off = O.gen_reads(5, 30000, 6, 3000, 0.0, False)
    short = np.ascontiguousarray(bases[:off[3]])
    # every read shorter than min_len (1000): the read set is empty
    cut_off = np.array([0, 400, 900, 1300], dtype=np.int64)
    orun = _run_arrays(short[:1300], cut_off)
    assert orun.paf == ""
    for n in (1, 2):
        _run_arrays(np.ascontiguousarray(bases[:off[n]]), np.ascontiguousarray(off[:n + 1]))


@pytest.mark.parametrize("slots", [1, 2])
def test_overlap_identical_and_repetitive_reads(slots):
    """Collisions: many copies of the same read plus tandem repeats — every query matches every target with many
    equally good chains (long open-chain lists, the chain kernel's lds tier, capacity rules)."""
    rng = np.random.default_rng(3)
    unit = "".join("ACGT"[i] for i in rng.integers(0, 4, 700))
    base_read = "".join("ACGT"[i] for i in rng.integers(0, 4, 3000))
    reads = [base_read] * 25 + [unit * 5] * 10 + [base_read[500:] + unit] * 5
    other, ooff = O.gen_reads(8, 40000, 60, 3000, 0.01, True)
    texts = reads + [other[ooff[i]:ooff[i + 1]].tobytes().decode() for i in range(60)]
    bases = np.frombuffer("".join(texts).encode(), dtype=np.uint8)
    off = np.cumsum([0] + [len(t) for t in texts]).astype(np.int64)
    _run_arrays(bases, off, slots=slots)


def test_overlap_non_acgt_letters():
    """IUPAC / lower-case letters go through the same 2-bit code as in the reference (sequence.go:59): no special casing."""
    bases, off = O.gen_reads(21, 60000, 150, 3000, 0.0, False)
    b = bases.copy()
    rng = np.random.default_rng(4)
    pos = rng.integers(0, len(b), 3000)
    b[pos] = np.frombuffer(b"NRYKMSWnacgt", dtype=np.uint8)[rng.integers(0, 12, 3000)]
    _run_arrays(b, off)


@pytest.mark.parametrize("tune", ["no_planner_thread=1", "host_select=1"])
def test_overlap_planner_variants(tune):
    """The planner without its own thread (executor slots extend the plan chain themselves, one at a time) and with the
    seed selection on the host threads instead of dp_select_seeds (DP_TUNE tokens)."""
    os.environ["DP_TUNE"] = tune
    try:
        _run_both(32, 60000, 500, 1500, 10, variable=True, slots=3)
        _run_both(113, 1500000, 3000, 10000, 13, max_rounds=3, slots=2)
    finally:
        del os.environ["DP_TUNE"]


@pytest.mark.parametrize("world,slots,seed,G,N,L,variable", [(2, 2, 31, 100000, 400, 5000, False), (4, 2, 31, 100000, 400, 5000, False),
                                                              (3, 2, 32, 60000, 500, 1500, True), (4, 1, 9, 500000, 2000, 5000, False)])
def test_round_pipelined_ranks_match_oracle(world, slots, seed, G, N, L, variable):
    """The pipelined round-parallel mode with `world` simulated ranks on one GPU: every rank's executor pipeline works on its
    residue class of rounds, one round per rank is exchanged per superstep and committed in order with the speculation
    check (the third case flags reads as ignored, so rounds get rejected and re-executed by their owners)."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(seed, G, N, L, 0.0, variable)
    kw = dict(k=10, seed_batch_size=1500) if seed == 9 else dict(k=10)
    max_rounds = 40 if seed == 9 else -1
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], rank=r, world=world, mode="round", slots=slots, **kw) for r in range(world)]
    committed, supersteps = 0, 0
    while not pipes[0].finished() and (max_rounds < 0 or committed < max_rounds):
        blobs = [pipes[r].wait_owned_blob() for r in range(world)]
        cs = [pipes[r].commit_gathered(blobs) for r in range(world)]
        assert len(set(cs)) == 1
        committed += cs[0]
        supersteps += 1
        assert supersteps < 5000
    rs = O.ReadSet(bases, off, min_len=1000)
    orun = O.OverlapRun(rs, max_rounds=committed if max_rounds >= 0 else -1, **kw)
    assert orun.rounds == committed
    for r in range(world):
        assert first_diff(pipes[r].all_paf(), orun.paf) is None
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        pipes[r].close()
    if variable:
        assert rs.ignore().sum() > 0


@pytest.mark.parametrize("mode", ["1", "0"])
def test_overlap_kmer_index_mode(mode):
    """DP_SCAN_INDEX=1: counts and segments come from the resident k-mer position index (dp_kindex.hip) instead of the
    scan kernels; =0 forbids it.  Same PAF, same ignore flags either way - whole runs, concurrent slots, the len%4==0
    top-level quirk (himem=false), the noisy k=13 regime and a non-default query type."""
    os.environ["DP_SCAN_INDEX"] = mode
    try:
        _, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=3)
        assert (st["idx_rounds"] > 0) == (mode == "1")
        _run_both(7, 100000, 300, 4000, 10, himem=False, max_rounds=3)
        _, st = _run_both(114, 1200000, 2000, 12000, 13, 0.002, True, max_rounds=4, slots=2)
        assert (st["idx_rounds"] > 0) == (mode == "1")
        assert (st["idx_hits"] > 0) == (mode == "1")
        _run_both(41, 100000, 400, 5000, 10, max_rounds=3, slots=2, query_type=4)
    finally:
        del os.environ["DP_SCAN_INDEX"]


@pytest.mark.parametrize("env", [{"DP_KX_BINS": "0"}, {"DP_KX_BINS_CAP": "300"}, {"DP_KX_BINS_CAP": "40"}, {"DP_KX_ONESHOT": "0"}, {"DP_KX_FUSE": "0"},
                                 {"DP_KX_DENSE": "1"}, {"DP_KX_DENSE": "1", "DP_KX_BINS_CAP": "300"}, {"DP_KX_DENSE": "1", "DP_KX_BINS_CAP": "40"}])
def test_kmer_index_counting_step_variants(monkeypatch, env):
    """The counting step of an index-mode round (dp_kindex.hip): hits binned by read range and counted in LDS (round 5, the default),
    round 4's hit records with one atomic per hit (DP_KX_BINS=0), bins that overflow - a workgroup's share that does not fit is
    counted the old way, what it reserved is marked unwritten, the fill pass walks the buckets (DP_KX_BINS_CAP: some bins at 300,
    every bin at 40) - the two-wait form (DP_KX_ONESHOT=0), the count in a launch of its own instead of inside kidx_offsets (DP_KX_FUSE=0) and
    the dense regime's small bins filled and sorted in LDS with batched reservations (round 6, DP_KX_DENSE=1 forces it on these small
    inputs; with bins that overflow too).  All read per call, so a variant set here is the variant that runs.  Same PAF, same ignore flags as the oracle: dense seeds (k = 10, hundreds
    of hits per read), sparse seeds (k = 13), reads that get flagged, five slots."""
    monkeypatch.setenv("DP_SCAN_INDEX", "1")
    for kk, vv in env.items():
        monkeypatch.setenv(kk, vv)
    _, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=3)
    assert st["idx_rounds"] > 0
    _, st = _run_both(114, 1200000, 2000, 12000, 13, 0.002, True, max_rounds=4, slots=5)
    assert st["idx_rounds"] > 0 and st["idx_hits"] > 0
    _run_both(7, 100000, 300, 4000, 10, himem=False, max_rounds=3)


@pytest.mark.parametrize("env", [{"DP_KINDEX_WIDE": "1"}, {"DP_TUNE": "kb_b1=9"}, {"DP_KB_MIN_PBITS": "22"}, {"DP_KB_MIN_PBITS": "30"}, {"DP_TUNE": "kindex_atomic=1"}])
def test_kmer_index_entry_formats(monkeypatch, env):
    """The resident k-mer position index stores an entry in 4, 5 or 8 bytes (dp_kbuild.hip): every format (reached on small inputs
    through DP_KB_MIN_PBITS, which only wastes bits), another width of the first pass, the all-eight-bytes build and the atomic
    scatter build give the oracle's PAF."""
    monkeypatch.setenv("DP_SCAN_INDEX", "1")
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    _, st = _run_both(114, 1200000, 2000, 12000, 13, 0.002, True, max_rounds=4, slots=2)
    assert st["idx_rounds"] > 0 and st["idx_hits"] > 0
    _, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=2)
    assert st["idx_rounds"] > 0


def test_overlap_kmer_index_unavailable_falls_back_to_scan():
    """k above the direct-addressed table's limit (14; lowered here through DP_KINDEX_MAX_K) or too little free HBM: the
    index reports itself unavailable and the rounds scan."""
    os.environ["DP_SCAN_INDEX"] = "1"
    os.environ["DP_KINDEX_MAX_K"] = "9"
    try:
        _, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=2)
        assert st["idx_rounds"] == 0
    finally:
        del os.environ["DP_SCAN_INDEX"]
        del os.environ["DP_KINDEX_MAX_K"]


def test_overlap_slots_with_ignores():
    """Short reads get flagged as ignored by earlier rounds: concurrent slots must discard invalidated speculation."""
    orun, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=3)
    assert orun.rounds >= 2


@pytest.mark.parametrize("lanes", [1, 2, 4])
@pytest.mark.parametrize("case", ["flags", "crowded", "plain"])
def test_overlap_planner_lanes(monkeypatch, lanes, case):
    """Planner lanes compute consecutive plans concurrently, each from a guess of where the plan before it ends (the seed budget
    is tested once per read, overlap.go:57-60; the guess counts the seeds of the windows' cached selections).  A wrong guess must
    cost a recomputation, never a different plan: "crowded" (k = 8: a sixth of all k-mers are seeds, nearly every window is
    re-selected and reverse complements collide) makes guesses fail, "flags" has rounds that flag reads."""
    import ctypes as C
    from downpore_amd.overlap import load_host
    monkeypatch.setenv("DPH_PLAN_LANES", str(lanes))
    H = load_host()
    H.dph_planner_counter.restype = C.c_int64
    H.dph_planner_counter.argtypes = [C.c_int]
    before = [H.dph_planner_counter(i) for i in range(3)]
    if case == "flags":
        orun, st = _run_both(32, 60000, 500, 1500, 10, variable=True, slots=3)
    elif case == "crowded":
        orun, st = _run_both(41, 200000, 600, 4000, 8, slots=3, seed_batch_size=1500)
    else:
        orun, st = _run_both(42, 400000, 1200, 5000, 12, slots=4, seed_batch_size=1000)
    after = [H.dph_planner_counter(i) for i in range(3)]
    print("lanes %d %s: rounds %d, plans computed %d, thrown away %d, erased %d" % ((lanes, case, orun.rounds) + tuple(a - b for a, b in zip(after, before))))
    assert orun.rounds >= 2


@pytest.mark.parametrize("slots", [1, 3])
def test_overlap_jobs_in_a_row_on_one_handle(slots):
    """reset() ends a job and keeps the handle - and with it the executor slots' and the planner's device contexts (streams,
    buffers, the flags and read items they hold) - for the next job on the same resident reads: three jobs in a row, the read
    set has rounds that flag reads (the contexts remember flag epochs: a new job's must not look like the old one's), every
    job's PAF and flags equal the oracle's."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(32, 60000, 500, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    orun = O.OverlapRun(rs, k=10)
    assert rs.ignore().sum() > 0 and orun.rounds >= 2
    reads = Reads(bases, off, min_len=1000)
    pipe = OverlapPipeline(reads, k=10, slots=slots, defer_init=True)
    for job in range(3):
        pipe.init()
        rounds = pipe.run()
        assert rounds == orun.rounds, "job %d" % job
        assert first_diff(pipe.all_paf(), orun.paf) is None, "job %d" % job
        assert np.array_equal(reads.ignore(), rs.ignore()), "job %d" % job
        pipe.reset()
        assert reads.ignore().sum() == 0
    pipe.close()


def test_overlap_himem_false_top_level_reads():
    """himem=false: reads are re-read as top-level sequences, len%4==0 scan quirk included."""
    _run_both(7, 100000, 300, 4000, 10, himem=False, max_rounds=3)


def test_values_table_matches_oracle():
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(3, 200000, 500, 4000, 0.01, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = rs.kmer_values(10)
    reads = Reads(bases, off, min_len=1000)
    pipe = OverlapPipeline(reads, k=10)
    assert np.array_equal(pipe.values(), want)
    pipe.close()


def test_cli_matches_oracle_cli(tmp_path):
    bases, off = O.gen_reads(9, 120000, 400, 5000, 0.0, False)
    fa = str(tmp_path / "reads.fa")
    O.write_fasta(fa, bases, off)
    a = subprocess.run([os.path.join(ROOT, "downpore_amd", "bin", "downpore"), "overlap", "-input", fa, "-k", "10"],
                       capture_output=True, check=True)
    b = subprocess.run([os.path.join(ROOT, "oracle", "_build", "dp_oracle"), "overlap", "-input", fa, "-k", "10"],
                       capture_output=True, check=True)
    assert first_diff(a.stdout.decode(), b.stdout.decode()) is None and len(a.stdout) > 0
    # aliases: -i / -k (commands/command.go:26-54)
    c = subprocess.run([os.path.join(ROOT, "downpore_amd", "bin", "downpore"), "overlap", "-i", fa, "--k", "10"],
                       capture_output=True, check=True)
    assert first_diff(c.stdout.decode(), a.stdout.decode()) is None


@pytest.mark.parametrize("world,seed,G,N,L,variable", [(4, 31, 100000, 400, 5000, False), (3, 32, 60000, 500, 1500, True)])
def test_round_parallel_protocol_matches_oracle(world, seed, G, N, L, variable):
    """Round-parallel multi-GPU mode, simulated with `world` contexts on one GPU: rank r executes round base+r
    speculatively, results are exchanged and committed in order with the speculation check.  The second case has many
    reads <= 2*overlap_size, so rounds DO flag reads as ignored and later speculative rounds must be discarded."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(seed, G, N, L, 0.0, variable)
    rs = O.ReadSet(bases, off, min_len=1000)
    orun = O.OverlapRun(rs, k=10)
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, rank=r, world=world, mode="round-batch") for r in range(world)]
    supersteps = 0
    short_commits = 0
    while not pipes[0].finished():
        base = pipes[0].committed_rounds()
        blobs = [pipes[r].exec_round_blob(base + r) for r in range(world)]
        cs = [pipes[r].commit_blobs(blobs) for r in range(world)]
        assert len(set(cs)) == 1
        if 0 < cs[0] < world and not pipes[0].finished():
            short_commits += 1
        supersteps += 1
        assert supersteps < 10 * orun.rounds + 10
    for r in range(world):
        assert first_diff(pipes[r].all_paf(), orun.paf) is None
        assert pipes[r].committed_rounds() == orun.rounds
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        pipes[r].close()
    if variable:
        assert rs.ignore().sum() > 0


@pytest.mark.parametrize("env", [{}, {"DP_DEVICE_CHUNK": "0"}, {"DP_TUNE": "cons_flag_every=3"}, {"DP_SCAN_INDEX": "1", "DP_TUNE": "cons_flag_every=2"},
                                 {"DP_SCAN_INDEX": "1"}, {"DP_SCAN_INDEX": "1", "DP_KX_ONESHOT": "0"}, {"DP_SCAN_INDEX": "1", "DPH_PRECHAIN": "0"},
                                 {"DP_SCAN_INDEX": "1", "DPH_PRECHAIN": "1", "DP_TUNE": "cons_flag_every=3"}, {"DP_INDEX_FILL_ROWS": "1"},
                                 {"DP_INDEX_FILL_ROWS": "1", "DP_SCAN_INDEX": "1"}])
@pytest.mark.parametrize("L,k,e", [(2500, 10, 0.0), (30000, 10, 0.01)])
def test_overlap_chunks_made_on_the_device(monkeypatch, env, L, k, e):
    """chunkWorker (overlap.go:253-318) runs on the device (dp_index_build_chunked): the survivors' segments stay in the scan
    buffer, the chunks and their {read, length, offset, inset} never visit the host, the consensus takes them from there.
    Short reads go in whole, 30 kb reads at k = 10 are cut into several chunks with the back-up of overlap / 2 and the
    150-seed tail rule.  Variants: host chunking (DP_DEVICE_CHUNK=0), the k-mer index or the scan kernels as producer, and
    every 2nd / 3rd window handed to the host consensus path (DP_TUNE=cons_flag_every=n), which then fetches chunks and segments
    after all.  Round 4: the index step in one go (hit records; DP_KX_ONESHOT=0: two waits as before) and the chunk stage
    launched behind the un-waited scan (DPH_PRECHAIN=0 / 1).  Same PAF as the oracle everywhere."""
    for kk, v in env.items():
        monkeypatch.setenv(kk, v)
    bases, off = O.gen_reads(41 + L, 120000, 400 if L < 10000 else 120, L, e, True)
    _run_arrays(bases, off, k=k, slots=2)


def test_scan_shard_exchange_inside_the_library():
    """The north-star multi-GPU layout through the C ABI's own entry points: ranks own contiguous read ranges,
    scan them with the HIP kernels, and dp_allgather_survivors exchanges the survivors device to device (no host hop).  Two
    in-process ranks on one GPU (dp_comm_init_local), with reads short enough that rounds flag reads as ignored, and the
    1-rank RCCL communicator (dp_comm_init) on the same job; every rank must print the oracle's PAF."""
    import threading
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(32, 60000, 500, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10)
    world = 2
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, rank=r, world=world, mode="scan-shard", comm="local") for r in range(world)]
    OverlapPipeline.link_local(pipes)
    errs = []

    def run(p):
        try:
            p.run()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(p,)) for p in pipes]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(world):
        d = first_diff(pipes[r].all_paf(), want.paf)
        assert d is None, (r, d)
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        pipes[r].close()
    assert want.paf.count("\n") > 100 and rs.ignore().sum() > 0
    # RCCL flavour, world of one (a 1-GPU box): same entry points, same PAF
    r1 = Reads(bases, off, min_len=1000)
    p1 = OverlapPipeline(r1, k=10, rank=0, world=1, mode="scan-shard", comm="rccl")
    p1.run()
    assert first_diff(p1.all_paf(), want.paf) is None
    p1.close()


@pytest.mark.parametrize("slots", [1, 3])
def test_scan_shard_rank_failure_before_the_exchange_does_not_hang_its_peers(monkeypatch, slots):
    """The same for a rank whose round fails BEFORE it reaches the exchange (dp_round_begin out of memory, a failed plan): the
    slot's communicator is aborted on every non-zero return of a sharded round, not only on the exchange's own failure."""
    import threading
    from downpore_amd.hip import DpError
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(32, 60000, 500, 1500, 0.0, True)
    world = 2
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, rank=r, world=world, mode="scan-shard", comm="local", slots=slots) for r in range(world)]
    OverlapPipeline.link_local(pipes)
    monkeypatch.setenv("DP_TUNE", "fail_begin_rank=1")
    errs = [None] * world

    def run(r):
        try:
            pipes[r].run()
        except DpError as e:
            errs[r] = e
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for a peer that failed before the exchange"
    assert all(e is not None for e in errs), errs
    assert "injected failure before the exchange" in str(errs[1])
    monkeypatch.delenv("DP_TUNE")
    for p in pipes:
        p.close()


@pytest.mark.parametrize("slots", [1, 3])
def test_scan_shard_rank_failure_does_not_hang_its_peers(monkeypatch, slots):
    """A rank that fails inside (or before) the survivor exchange must not leave the others waiting for it: the in-process
    group is marked failed, every rank's step returns an error (dp_allgather_survivors / dp_comm_abort)."""
    import threading
    from downpore_amd.hip import DpError
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(32, 60000, 500, 1500, 0.0, True)
    world = 2
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, rank=r, world=world, mode="scan-shard", comm="local", slots=slots) for r in range(world)]
    OverlapPipeline.link_local(pipes)
    monkeypatch.setenv("DP_TUNE", "comm_fail_rank=1")
    errs = [None] * world

    def run(r):
        try:
            pipes[r].run()
        except DpError as e:
            errs[r] = e
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "a rank is still waiting for a peer that failed"
    assert all(e is not None for e in errs), errs
    assert "injected failure" in str(errs[1])
    monkeypatch.delenv("DP_TUNE")
    for p in pipes:
        p.close()


@pytest.mark.parametrize("shard_queries", ["1", "0"])
def test_scan_shard_with_executor_slots(monkeypatch, shard_queries):
    """(shard_queries: the round's query windows dealt to the ranks as well - every rank chains and builds the consensus for its
    own share, the ranks' PAF text and SetIgnore ids are all-gathered and joined in rank order - or every rank doing every window.)
    Scan-shard mode keeps its executor slots: a step runs `slots` consecutive rounds concurrently, slot i exchanging its
    survivors on its own communicator, and commits them in order with the speculation check (rounds that meet a read flagged by
    an earlier round of the same batch are run again).  Two in-process ranks x three slots on one GPU, on reads short enough
    that many rounds flag reads; a 1-rank RCCL job with two slots.  Every rank must print the oracle's PAF."""
    import threading
    from downpore_amd.overlap import OverlapPipeline, Reads
    monkeypatch.setenv("DP_TUNE", "no_shard_queries=%d" % (1 if shard_queries == "0" else 0))
    bases, off = O.gen_reads(33, 60000, 700, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10, seed_batch_size=1500)
    assert want.rounds >= 8 and rs.ignore().sum() > 0, (want.rounds, int(rs.ignore().sum()))
    world = 2
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, seed_batch_size=1500, rank=r, world=world, mode="scan-shard", comm="local", slots=3)
             for r in range(world)]
    OverlapPipeline.link_local(pipes)
    errs = []

    def run(p):
        try:
            p.run()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(p,)) for p in pipes]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(world):
        d = first_diff(pipes[r].all_paf(), want.paf)
        assert d is None, (r, d)
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        assert pipes[r].committed_rounds() == want.rounds
        pipes[r].close()
    r1 = Reads(bases, off, min_len=1000)
    p1 = OverlapPipeline(r1, k=10, seed_batch_size=1500, rank=0, world=1, mode="scan-shard", comm="rccl", slots=2)
    p1.run()
    assert first_diff(p1.all_paf(), want.paf) is None
    p1.close()


def _fastq_quals(bases, seed):
    """Quality characters with structure: low-quality stretches, so that the weighting changes which k-mers win."""
    rng = np.random.default_rng(seed)
    q = rng.integers(35, 74, len(bases)).astype(np.uint8)  # '#'..'I'
    for _ in range(len(bases) // 700):
        a = int(rng.integers(0, max(1, len(bases) - 300)))
        q[a:a + int(rng.integers(20, 300))] = 33 + int(rng.integers(0, 4))  # phred 0..3
    return q


def test_overlap_fastq_quality_weighted_seeds(tmp_path, monkeypatch):
    """FASTQ input: AddSeeds weights every k-mer's value by the quality byte of its middle base
    (seeds/seeds.go:99-101).  The oracle and the product (device selection through the window cache, host re-selection of
    touched windows; and the per-plan dp_select_seeds path) must print the same PAF - and a different one than without
    qualities, or the test would prove nothing."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(77, 90000, 300, 4000, 0.0, True)
    quals = _fastq_quals(bases, 5)
    rs = O.ReadSet(bases, off, min_len=1000, quals=quals)
    want = O.OverlapRun(rs, k=10)
    plain = O.OverlapRun(O.ReadSet(bases, off, min_len=1000), k=10)
    assert want.paf != plain.paf and want.paf.count("\n") > 200
    for cache in ("1", "0"):
        monkeypatch.setenv("DP_WINDOW_CACHE", cache)
        reads = Reads(bases, off, min_len=1000, quals=quals)
        pipe = OverlapPipeline(reads, k=10, slots=3)
        pipe.run()
        d = first_diff(pipe.all_paf(), want.paf)
        assert d is None, (cache, d)
        assert np.array_equal(reads.ignore(), rs.ignore())
        pipe.close()
    monkeypatch.delenv("DP_WINDOW_CACHE")
    # a second and third job on the same handle (reset() keeps the slots' and the planner's contexts, which borrow the reads:
    # the quality bytes were uploaded once with the reads and must still weight the selection)
    reads = Reads(bases, off, min_len=1000, quals=quals)
    pipe = OverlapPipeline(reads, k=10, slots=3, defer_init=True)
    for job in range(3):
        pipe.init()
        pipe.run()
        d = first_diff(pipe.all_paf(), want.paf)
        assert d is None, (job, d)
        pipe.reset()
    pipe.close()
    # the same through the FASTQ reader and the CLI of both sides
    fq = str(tmp_path / "reads.fq")
    O.write_fastq(fq, bases, off, quals)
    a = subprocess.run([os.path.join(ROOT, "downpore_amd", "bin", "downpore"), "overlap", "-input", fq, "-k", "10"],
                       capture_output=True, check=True)
    b = subprocess.run([os.path.join(ROOT, "oracle", "_build", "dp_oracle"), "overlap", "-input", fq, "-k", "10"],
                       capture_output=True, check=True)
    assert first_diff(a.stdout.decode(), b.stdout.decode()) is None
    assert first_diff(a.stdout.decode(), want.paf) is None


def test_whole_job_through_the_published_host_header_only():
    """The pipeline bench.py measures has a published boundary (include/downpore_host.h; Go: integration/gpuhost +
    integration/commands/gpu_overlap.go).  This drives a whole job with raw ctypes calls of entry points that header declares -
    and of nothing else - the way the Go command does: reads, open, init, step / round_paf until done; PAF and stderr lines
    against the oracle."""
    import ctypes as C
    import re
    from downpore_amd.overlap import host_lib_path, load_host
    load_host()
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "downpore_host.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(dph_[a-z_0-9]+)\s*\(", txt))

    class HeaderOnly:
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            assert name in declared, "%s is not declared in include/downpore_host.h" % name
            return getattr(self._lib, name)
    H = HeaderOnly(C.CDLL(host_lib_path()))
    vp = C.c_void_p
    bases, off = O.gen_reads(1, 250000, 1000, 5000, 0.0, False)  # BASELINE config 1, at the command's default k
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10)
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    o = np.ascontiguousarray(off, dtype=np.int64)
    H.dph_reads_from_arrays.restype = vp
    reads = H.dph_reads_from_arrays(C.c_void_p(b.ctypes.data), C.c_void_p(o.ctypes.data), C.c_int64(len(o) - 1), C.c_int64(1000), C.c_int(1))
    assert reads
    H.dph_overlap_open.restype = vp
    h = H.dph_overlap_open(vp(reads), C.c_int(0))
    assert h
    params = np.array([1000, 10, 15, 10000, 10000, 20000, 1 | (1 << 8), 8], dtype=np.int64)
    assert H.dph_overlap_init(vp(h), C.c_void_p(params.ctypes.data), C.c_double(0.25), None) == 0
    H.dph_overlap_round_paf.restype = C.POINTER(C.c_char)
    H.dph_overlap_errtext.restype = C.POINTER(C.c_char)
    H.dph_overlap_step_lines.restype = C.c_int64
    paf, rounds, lines = [], 0, 0
    while True:
        c = H.dph_overlap_step(vp(h))
        assert c >= 0
        if c == 0:
            break
        rounds += c
        n = C.c_int64(0)
        p = H.dph_overlap_round_paf(vp(h), C.byref(n))
        paf.append(C.string_at(p, n.value).decode())
        lines += H.dph_overlap_step_lines(vp(h))
    n = C.c_int64(0)
    p = H.dph_overlap_errtext(vp(h), C.byref(n))
    err = C.string_at(p, n.value).decode()
    got = "".join(paf)
    assert rounds == want.rounds and H.dph_overlap_done(vp(h)) == 1
    assert first_diff(got, want.paf) is None and lines == got.count("\n") > 1000
    assert err.startswith("Counting all 10-mers in the input...") and err.count("\nTotal ") + err.count("sequences.Total ") >= 1
    assert err.count(" hits across ") == rounds
    ig = np.zeros(len(o) - 1, dtype=np.uint8)
    H.dph_reads_get_ignore(vp(reads), C.c_void_p(ig.ctypes.data))
    assert np.array_equal(ig, rs.ignore())
    H.dph_overlap_destroy(vp(h))
    H.dph_reads_free(vp(reads))


def test_round_parallel_exchange_inside_the_library():
    """The round-parallel layout with its result exchange inside the C ABI (dp_allgather_blobs behind dph_overlap_superstep): two
    in-process ranks x three slots on one GPU (host copies between the handles) on reads that flag reads, and the 1-rank RCCL
    communicator; every rank must print the oracle's PAF."""
    import threading
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(33, 60000, 700, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10, seed_batch_size=1500)
    assert want.rounds >= 8 and rs.ignore().sum() > 0
    world = 2
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, seed_batch_size=1500, rank=r, world=world, mode="round", comm="local", slots=3)
             for r in range(world)]
    OverlapPipeline.link_local(pipes)
    errs = []

    def run(p):
        try:
            p.run()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(p,)) for p in pipes]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for r in range(world):
        d = first_diff(pipes[r].all_paf(), want.paf)
        assert d is None, (r, d)
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        pipes[r].close()
    r1 = Reads(bases, off, min_len=1000)
    p1 = OverlapPipeline(r1, k=10, seed_batch_size=1500, rank=0, world=1, mode="round", comm="rccl", slots=2)
    p1.run()
    assert first_diff(p1.all_paf(), want.paf) is None
    p1.close()


@pytest.mark.parametrize("world,k,env", [(2, 10, {}), (3, 13, {}), (4, 11, {"DP_KB_MIN_PBITS": "30"}), (3, 10, {"DP_KINDEX_WIDE": "1"})])
def test_kmer_index_built_in_shares_and_all_gathered(monkeypatch, world, k, env):
    """BASELINE.json's "RCCL all-gather of the seed index", for the resident k-mer position index (round 5): with a communicator
    announced (dp_kindex_set_comm) every rank radix-sorts the k-mers of 1 / world of the first-digit buckets and the shares - entries,
    bucket offsets, k-mer counts - are all-gathered in place (here: contexts of one process on one GPU, device-to-device copies; across
    processes the same call sequence runs grouped ncclBroadcasts).  Every rank's gathered index must be the single-rank build's:
    same entry count, same bucket starts, same entries per bucket (order inside a bucket is free) - in the 4-byte, 5-byte and 8-byte
    entry formats - and the k-mer value table computed from the gathered counts must be bit-identical."""
    import ctypes as C
    import threading
    import downpore_amd
    from downpore_amd import hip
    monkeypatch.setenv("DP_SCAN_INDEX", "1")
    for kk, vv in env.items():
        monkeypatch.setenv(kk, vv)
    L = hip.load_library()
    L.dp_comm_init_local.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.dp_kindex_set_comm.argtypes = [C.c_void_p, C.c_void_p]
    L.dp_kindex_digest.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.dp_comm_destroy.argtypes = [C.c_void_p]
    bases, off = O.gen_reads(90 + k, 300000, 900, 4000, 0.01, True)
    ref = downpore_amd.Context(0)
    ref.upload_reads(bases, off)
    ref.scan_prepare(k)
    want = np.zeros(3, dtype=np.uint64)
    assert L.dp_kindex_digest(ref.h, k, want.ctypes.data) == 0 and want[0] > 3000000
    want_values = ref.kmer_values(k)
    ctxs = [downpore_amd.Context(0) for _ in range(world)]
    for c in ctxs:
        c.upload_reads(bases, off)
    hs = (C.c_void_p * world)(*[c.h for c in ctxs])
    comms = (C.c_void_p * world)()
    assert L.dp_comm_init_local(hs, world, comms) == 0
    for r in range(world):
        assert L.dp_kindex_set_comm(ctxs[r].h, comms[r]) == 0
    errs = []

    def build(c):
        try:
            c.scan_prepare(k)  # collective
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=build, args=(c,)) for c in ctxs]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for r in range(world):
        got = np.zeros(3, dtype=np.uint64)
        assert L.dp_kindex_digest(ctxs[r].h, k, got.ctypes.data) == 0
        assert got.tolist() == want.tolist(), (r, got, want)
        v = ctxs[r].kmer_values(k)
        assert np.array_equal(v.view(np.uint64), want_values.view(np.uint64)), r
    for r in range(world):
        L.dp_kindex_set_comm(ctxs[r].h, None)
        L.dp_comm_destroy(comms[r])
        ctxs[r].close()
    ref.close()


def test_index_share_gather_through_a_real_rccl_communicator(monkeypatch):
    """DP_KINDEX_SHARD=force: the share / gather path with a communicator of ONE rank made by ncclCommInitRank - the only way a one-GPU
    box can put the RCCL flavour of dp_comm_allgather_ranges (grouped ncclBroadcasts between device buffers, out of place for the
    entries, in place for offsets and counts) and of dp_gather_blobs through a real librccl.  Whole job, PAF against the oracle."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    monkeypatch.setenv("DP_SCAN_INDEX", "1")
    monkeypatch.setenv("DP_KINDEX_SHARD", "force")
    bases, off = O.gen_reads(33, 60000, 700, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10, seed_batch_size=1500)
    r1 = Reads(bases, off, min_len=1000)
    p1 = OverlapPipeline(r1, k=10, seed_batch_size=1500, rank=0, world=1, mode="round", comm="rccl", slots=2)
    p1.text_root(0)
    p1.run()
    assert first_diff(p1.all_paf(), want.paf) is None
    assert p1.stats_total()["idx_rounds"] > 0
    p1.close()


def test_round_parallel_job_on_an_index_built_in_shares():
    """The whole round-parallel job (three in-process ranks x two slots, reads that flag reads) with the k-mer position index built in
    shares and all-gathered before the first round: dph_overlap_init is then collective.  Rank 0 prints the oracle's PAF."""
    import threading
    from downpore_amd.overlap import OverlapPipeline, Reads
    os.environ["DP_SCAN_INDEX"] = "1"
    try:
        bases, off = O.gen_reads(33, 60000, 700, 1500, 0.0, True)
        rs = O.ReadSet(bases, off, min_len=1000)
        want = O.OverlapRun(rs, k=10, seed_batch_size=1500)
        world = 3
        readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
        pipes = [OverlapPipeline(readsets[r], k=10, seed_batch_size=1500, rank=r, world=world, mode="round", comm="local", slots=2, defer_init=True)
                 for r in range(world)]
        OverlapPipeline.link_local(pipes)  # (before init: the index of this job is built in shares over the communicator)
        for r, p in enumerate(pipes):
            p.text_root(0)
            p.keep_text(r == 0)
        errs = []

        def run(p):
            try:
                p.init()
                p.run()
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=run, args=(p,)) for p in pipes]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not errs, errs
        d = first_diff(pipes[0].all_paf(), want.paf)
        assert d is None, d
        for r in range(world):
            st = pipes[r].stats_total()
            assert st["idx_rounds"] > 0
            assert pipes[r].committed_rounds() == want.rounds
            assert np.array_equal(readsets[r].ignore(), rs.ignore())
            pipes[r].close()
    finally:
        del os.environ["DP_SCAN_INDEX"]


def test_round_parallel_text_goes_to_the_printing_rank_only():
    """dph_overlap_text_root(0): a superstep all-gathers the rounds' control records (flags, read lists, counts) and gathers their PAF
    text to rank 0 alone (dp_gather_blobs).  Three in-process ranks x two slots on reads that flag reads: rank 0 prints the oracle's
    PAF, ranks 1 and 2 print nothing, every rank ends with the oracle's ignore flags and round count; then the same through a
    1-rank RCCL communicator (ncclSend / ncclRecv are never issued there, the root's own text never leaves the host)."""
    import threading
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(33, 60000, 700, 1500, 0.0, True)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=10, seed_batch_size=1500)
    assert want.rounds >= 8 and rs.ignore().sum() > 0
    world = 3
    readsets = [Reads(bases, off, min_len=1000) for _ in range(world)]
    pipes = [OverlapPipeline(readsets[r], k=10, seed_batch_size=1500, rank=r, world=world, mode="round", comm="local", slots=2)
             for r in range(world)]
    OverlapPipeline.link_local(pipes)
    for r, p in enumerate(pipes):
        p.text_root(0)
        p.keep_text(r == 0)
    errs = []

    def run(p):
        try:
            p.run()
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(p,)) for p in pipes]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    d = first_diff(pipes[0].all_paf(), want.paf)
    assert d is None, d
    for r in range(world):
        if r:
            assert pipes[r].all_paf() == ""
        assert pipes[r].committed_rounds() == want.rounds
        assert np.array_equal(readsets[r].ignore(), rs.ignore())
        pipes[r].close()
    r1 = Reads(bases, off, min_len=1000)
    p1 = OverlapPipeline(r1, k=10, seed_batch_size=1500, rank=0, world=1, mode="round", comm="rccl", slots=2)
    p1.text_root(0)
    p1.run()
    assert first_diff(p1.all_paf(), want.paf) is None
    p1.close()


@pytest.mark.parametrize("k,G,N,L,variable", [(10, 250000, 1000, 5000, False), (13, 3000000, 6000, 10000, False)])
def test_consensus_layouts_agree(monkeypatch, k, G, N, L, variable):
    """consensus_full_kernel runs in two LDS layouts (int16 small tier first, the large tier for the windows it lists).  The same
    jobs with the small tier switched off (DP_CONS_LAYOUTS=nosmall: every window in the large layout, round 2's kernel) must print the
    same PAF as with it - and as the oracle."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(11, G, N, L, 0.0, variable)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=k, max_rounds=4)
    got = {}
    # (small, huge): huge = "1" puts the 144 KB layout (round 4: the dense regime's windows) behind the large one from the first
    # round on ("0": never; unset: from the round after the first window that did not fit the large one)
    # lazy = "0": the large layout launched behind the small one in every round; default: only after the wait, for a round whose small
    # layout listed a window (and at once for the 64 rounds after such a round)
    for small, huge, lazy in (("1", "0", "1"), ("0", "0", "1"), ("1", "1", "1"), ("0", "1", "1"), ("1", "0", "0")):
        monkeypatch.setenv("DP_CONS_LAYOUTS", ",".join((["nosmall"] if small == "0" else []) + ["huge" if huge == "1" else "nohuge"] + (["eager"] if lazy == "0" else [])))
        pipe = OverlapPipeline(Reads(bases, off, min_len=1000), k=k, slots=3)
        pipe.run(4)
        got[(small, huge, lazy)] = pipe.all_paf()
        pipe.close()
        assert first_diff(got[(small, huge, lazy)], want.paf) is None, (small, huge, lazy)
    monkeypatch.delenv("DP_CONS_LAYOUTS")


@pytest.mark.parametrize("k,G,N,L,variable,err", [(10, 250000, 1000, 5000, False, 0.0), (13, 3000000, 6000, 10000, False, 0.0),
                                                   (13, 1500000, 3000, 8000, True, 0.03)])
def test_chain_shortcuts_agree(monkeypatch, k, G, N, L, variable, err):
    """The chaining kernels decide a 'perfect chain' (one chain from the first event that every event extends) for all events at
    once instead of walking them, and leave final chains in their scratch columns when the consumers are on the device.  The same
    jobs with the walk forced for every pair (DP_CHAIN_PERFECT=0) and with the chains packed (DP_CHAIN_PACK=1) must print the
    oracle's PAF as well - error-free reads (nearly every pair perfect) and reads with errors (many are not)."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    bases, off = O.gen_reads(17, G, N, L, err, variable)
    rs = O.ReadSet(bases, off, min_len=1000)
    want = O.OverlapRun(rs, k=k, max_rounds=4)
    # (prechain = the chunk stage launched behind the un-waited scan, "0" = after the wait; prestage "0": the query block is uploaded by
    # dp_find_overlaps itself)
    for perfect, pack, prestage, prechain in (("1", "0", "1", "1"), ("0", "0", "1", "1"), ("1", "1", "1", "1"), ("1", "0", "0", "0"), ("1", "0", "1", "0")):
        monkeypatch.setenv("DP_CHAIN_PERFECT", perfect)
        monkeypatch.setenv("DP_CHAIN_PACK", pack)
        monkeypatch.setenv("DPH_PRECHAIN", prechain)
        monkeypatch.setenv("DP_TUNE", "no_query_prestage=%d" % (1 if prestage == "0" else 0))
        pipe = OverlapPipeline(Reads(bases, off, min_len=1000), k=k, slots=3)
        pipe.run(4)
        got = pipe.all_paf()
        pipe.close()
        assert first_diff(got, want.paf) is None, (perfect, pack, prestage, prechain)
