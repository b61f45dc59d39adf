"""GPU parity of the `map` path (A18-A21): reference seeding + chunk index + window scans + Matches + SeedSequence.Match
chaining + mapper control flow must print the ORACLE's PAF line for line (read order)."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu


def first_diff(a, b):
    if a == b:
        return None
    la, lb = a.split("\n"), b.split("\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return "line %d:\n  got  %s\n  want %s" % (i, x, y)
    return "line counts differ: got %d want %d" % (len(la), len(lb))


def _case(seed, G, N, L, e, variable, circular, k=11, short_reads=False):
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off = O.gen_reads(seed, G, N, L, e, variable)
    if short_reads:  # append reads <= 2*query_size, some with len % 4 == 0 (top-level scan quirks on both strands)
        extra_b, extra_o = O.gen_reads(seed, G, 40, 1400, e, True)
        cut = [1200, 1600, 1996, 2000, 900, 1333]
        parts, offs = [bases], [off]
        b2, o2 = [], [0]
        for i in range(40):
            ln = min(int(extra_o[i + 1] - extra_o[i]), cut[i % len(cut)])
            b2.append(extra_b[extra_o[i]:extra_o[i] + ln])
            o2.append(o2[-1] + ln)
        bases = np.concatenate([bases] + b2)
        off = np.concatenate([off, off[-1] + np.array(o2[1:], dtype=np.int64)])
    oref = O.ReadSet(genome, goff, min_len=0, himem=False)
    oreads = O.ReadSet(bases, off, min_len=500, himem=False)
    want, werr = O.map_run(oref, oreads, circular=circular, k=k)
    ref = Reads(genome, goff, min_len=0, himem=False)
    reads = Reads(bases, off, min_len=500, himem=False)
    got, gerr, st = map_reads(ref, reads, circular=circular, k=k)
    d = first_diff(got, want)
    assert d is None, d
    assert gerr == werr
    return want, st


@pytest.mark.parametrize("seed,G,N,L,e,variable,circular", [(3, 200000, 300, 8000, 0.0, False, True),
                                                           (4, 150000, 300, 6000, 0.05, True, True),
                                                           (5, 300000, 200, 9000, 0.10, True, False)])
def test_map_paf_bit_exact(seed, G, N, L, e, variable, circular):
    want, st = _case(seed, G, N, L, e, variable, circular)
    assert want.count("\n") > N // 2
    assert st["n_windows"] >= N


def test_map_single_seeds_host_walk_agrees(monkeypatch):
    """AddSingleSeeds: the windows' best k-mers and their count regions' candidates come from the device, the host walks the windows in
    order and probes the candidates (round 4); DP_TUNE=map_seeds_host=1 does the whole walk on the host as before.  Same PAF as the oracle
    either way - linear and circular references, lengths with every len % 4 (the count region's skipBack)."""
    for G in (200001, 200002, 200003, 200004):
        _case(3, G, 60, 6000, 0.02, True, G % 2 == 0)
    monkeypatch.setenv("DP_TUNE", "map_seeds_host=1")
    _case(3, 200002, 60, 6000, 0.02, True, True)
    monkeypatch.delenv("DP_TUNE")


def test_map_reads_travel_packed_or_as_ascii(monkeypatch):
    """The mapper packs its reads on the host (AVX2 or scalar, the worker pool) and hands them to dp_reads_upload_packed_rc (round 6);
    DP_TUNE=map_ascii_upload=1 sends the ASCII and packs on the device as before.  The oracle's PAF every way - read lengths of every
    length mod 4, reads shorter than a window, a circular and a linear reference."""
    cases = [(4, 150001, 300, 5000, 0.05, True, True), (4, 150002, 300, 700, 0.02, True, False)]
    for tune in ("", "pack_scalar=1", "map_ascii_upload=1"):
        if tune:
            monkeypatch.setenv("DP_TUNE", tune)
        for c in cases:
            _case(*c, short_reads=True)
    monkeypatch.delenv("DP_TUNE")


def test_map_threads_reads_in_flight_and_parked_blocks(monkeypatch):
    """The mapper deals its reads to host threads (DP_MAP_THREADS, default 8 from 16 host cores up) that keep DP_MAP_INFLIGHT reads in flight each (one window
    per read and dp_map_windows call); the contexts of a finished run park their device and pinned blocks in the library's cache and the
    next run takes them from there (round 5).  Same PAF as the oracle with one thread, many small calls, few large ones, and across
    dp_release_device_caches() - which returns what was parked, then nothing."""
    import ctypes as C
    from downpore_amd.hip import load_library
    lib = load_library()
    lib.dp_release_device_caches.restype = C.c_int64
    args = (5, 300000, 2600, 9000, 0.10, True, False)
    monkeypatch.setenv("DP_TUNE", "map_min_reads_per_thread=300")   # (2 048 by default: these reads would all go to one thread)
    _case(*args)
    assert lib.dp_release_device_caches() > 0
    assert lib.dp_release_device_caches() == 0
    # the reads are dealt to the threads in blocks of 1 024 and travel to the device while the first ones are mapped (round 5,
    # dp_reads_upload_rc_begin with DP_TUNE=map_async_upload=1; 0: all of them first): one thread, more threads than blocks, windows per call
    for threads, inflight, asyn in (("1", "64", "1"), ("2", "500", "1"), ("4", "100000", "1"), ("3", "700", "0"), ("8", "2730", "1")):
        monkeypatch.setenv("DP_MAP_THREADS", threads)
        monkeypatch.setenv("DP_MAP_INFLIGHT", inflight)
        monkeypatch.setenv("DP_TUNE", "map_min_reads_per_thread=300,map_async_upload=%s" % asyn)
        _case(*args)
    from downpore_amd.overlap import load_host
    H = load_host()
    H.dph_release_caches.restype = C.c_int64
    assert H.dph_release_caches() > 0   # (the host library's call gives the device library's parked blocks back too)
    assert lib.dp_release_device_caches() == 0


def test_map_dynamic_match_on_one_lane_agrees(monkeypatch):
    """dynamicMatch probes the target's seeds with 64 lanes at once (round 4); DP_TUNE=map_one_lane=1 runs the reference's loop nest on
    one lane as before.  Both must print the oracle's PAF - reads with errors, so that chains break, ratchets move and several
    chains start from one query seed."""
    monkeypatch.setenv("DP_TUNE", "map_one_lane=1")
    _case(5, 300000, 200, 9000, 0.10, True, False)
    monkeypatch.delenv("DP_TUNE")
    _case(6, 250000, 250, 7000, 0.15, True, True)


@pytest.mark.parametrize("shards,G,k,e", [(3, 6000000, 11, 0.10), (4, 2600000, 9, 0.05), (2, 1500000, 13, 0.0)])
def test_map_reference_index_in_shards(monkeypatch, shards, G, k, e):
    """BASELINE config 5's layout on one GPU: the reference chunks are dealt to DP_MAP_SHARDS contexts in contiguous ranges
    (whole 64-chunk words), every shard answers Matches() for its own words with the sets' GLOBAL windows and counts
    (dp_index_set_global), and performMapping's ratchets travel from shard to shard, forward strand first
    (dp_map_windows_shard).  The PAF must be the oracle's - i.e. the unsharded result: k = 9 makes every seed frequent
    (long windows, the 16-ladder and its gather order), k = 13 makes them rare (short sets, early returns)."""
    monkeypatch.setenv("DP_MAP_SHARDS", str(shards))
    want, st = _case(11 + shards, G, 400, 7000, e, True, True, k=k)
    assert st["n_chunks"] > 64 * (shards - 1)
    assert want.count("\n") > 200


def test_map_short_reads_and_len_mod4_quirks():
    want, st = _case(6, 120000, 100, 5000, 0.02, True, True, short_reads=True)
    assert want.count("\n") > 50


def test_map_cli_matches_oracle_cli(tmp_path):
    """`downpore map -input reads.fa -reference ref.fa` (product CLI, reference flag names and aliases, commands/map.go:17-22)
    against the oracle's CLI on the same FASTA files."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = 150000
    genome = np.frombuffer(O.gen_genome(8, G), dtype=np.uint8)
    bases, off = O.gen_reads(8, G, 120, 6000, 0.03, True)
    ref_fa, reads_fa = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fa")
    O.write_fasta(ref_fa, genome, np.array([0, G], dtype=np.int64), prefix="chr")
    O.write_fasta(reads_fa, bases, off)
    prod = os.path.join(root, "downpore_amd", "bin", "downpore")
    orac = os.path.join(root, "oracle", "_build", "dp_oracle")
    a = subprocess.run([prod, "map", "-input", reads_fa, "-reference", ref_fa], capture_output=True, check=True)
    b = subprocess.run([orac, "map", "-input", reads_fa, "-reference", ref_fa], capture_output=True, check=True)
    assert first_diff(a.stdout.decode(), b.stdout.decode()) is None and a.stdout.count(b"\n") > 60
    c = subprocess.run([prod, "map", "-i", reads_fa, "-r", ref_fa, "-k", "11"], capture_output=True, check=True)
    assert first_diff(c.stdout.decode(), a.stdout.decode()) is None


def test_map_cli_fastq_reads(tmp_path):
    """`map` with the reads in a FASTQ file (qualities play no part in mapping, the parser does): product CLI == oracle CLI."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G = 120000
    genome = np.frombuffer(O.gen_genome(18, G), dtype=np.uint8)
    bases, off = O.gen_reads(18, G, 100, 5000, 0.04, True)
    quals = np.random.default_rng(2).integers(33, 74, len(bases)).astype(np.uint8)
    ref_fa, reads_fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    O.write_fasta(ref_fa, genome, np.array([0, G], dtype=np.int64), prefix="chr")
    O.write_fastq(reads_fq, bases, off, quals)
    prod = os.path.join(root, "downpore_amd", "bin", "downpore")
    orac = os.path.join(root, "oracle", "_build", "dp_oracle")
    a = subprocess.run([prod, "map", "-input", reads_fq, "-reference", ref_fa], capture_output=True, check=True)
    b = subprocess.run([orac, "map", "-input", reads_fq, "-reference", ref_fa], capture_output=True, check=True)
    assert first_diff(a.stdout.decode(), b.stdout.decode()) is None and a.stdout.count(b"\n") > 50
