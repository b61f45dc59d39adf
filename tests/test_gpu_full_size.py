"""GPU parity at the FULL size of the BASELINE configs (what the driver's `pytest -m gpu` run proves, not a builder-side
script): config 2 — every one of the 599 rounds of the 100k x 10 kb job against the SHA-256 the oracle alone produced
(tests/golden_full/config2.json, tools/make_golden_full.py); config 3 — 50k x 8 kb `map` against the oracle run right here
(1.6 s on one core); config 4 — the first rounds of the 1M x 10 kb set (10 Gbase resident, k-mer index 80 GB) against
the oracle's fixture for the same rounds."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden_full", name + ".json")))


def _overlap_against_fixture(g, slots, max_rounds=-1):
    from downpore_amd.overlap import OverlapPipeline, Reads
    gen = g["generator"]
    bases, off = O.gen_reads(gen["seed"], gen["genome"], gen["reads"], gen["read_len"], gen["error"], gen["variable"])
    reads = Reads(bases, off, min_len=1000)
    del bases
    pipe = OverlapPipeline(reads, device=0, k=g["k"], slots=slots)
    rounds = pipe.run(max_rounds)
    paf = pipe.all_paf()
    st = pipe.stats()
    pipe.close()
    assert rounds == g["rounds"]
    assert paf.count("\n") == g["paf_lines"]
    assert paf.split("\n")[:4] == g["paf_head"]
    assert hashlib.sha256(paf.encode()).hexdigest() == g["paf_sha256"]
    assert hashlib.sha256(reads.ignore().tobytes()).hexdigest() == g["ignore_sha256"]
    return st


def test_config2_whole_job_matches_oracle_fixture():
    """BASELINE config 2, all rounds, executor pipeline with 6 slots (what bench.py times)."""
    g = _golden("config2")
    assert g["rounds"] > 500 and g["paf_lines"] > 3000000
    st = _overlap_against_fixture(g, slots=6)
    assert st["idx_rounds"] == 1  # 1 Gbase: served by the resident k-mer position index


def test_config2_whole_job_scan_kernels(monkeypatch):
    """The same job with the scan kernels instead of the k-mer index (first 150 rounds would not tell the two modes apart
    less than all of them do, and the whole job takes a second)."""
    monkeypatch.setenv("DP_SCAN_INDEX", "0")
    st = _overlap_against_fixture(_golden("config2"), slots=4)
    assert st["idx_rounds"] == 0


@pytest.mark.parametrize("case,slots", [("config2_k10_e0_first_16_rounds", 1), ("config2_k10_e0_first_16_rounds", 5),
                                        ("config2_k10_e003_first_6_rounds", 1), ("config2_k10_e003_first_6_rounds", 5)])
def test_config2_dense_regime_matches_oracle_fixture(case, slots):
    """SURVEY 8(d)'s dense-seed regime at config 2's size (k = 10, the command's default): every read is indexed (~190 k chunks
    per round, W ~ 3 k words) and the index query streams ~210 MB of posting words per launch - the regime the north star's
    "HBM roofline during index-query" is quoted on (bench.py's index_query_dense leg runs these very rounds).  First rounds
    against the oracle's fixture, error-free and at 3 % errors, with one round in flight and with five."""
    g = _golden(case)
    assert g["k"] == 10 and g["paf_lines"] > 20000
    st = _overlap_against_fixture(g, slots=slots, max_rounds=g["max_rounds"])
    assert st["n_indexed"] > 100000  # (last round: the dense regime indexes every read, in pieces)


@pytest.mark.parametrize("case", ["config2_k13_e0002_first_24_rounds", "config2_k13_variable_first_24_rounds"])
def test_config2_k13_variants_match_oracle_fixture(case):
    """SURVEY 8(d)'s other k = 13 inputs at config 2's size: 0.2 % errors, and the L*U[0.5,1.5] length model; first 24 rounds
    against the oracle's fixture.  (Neither flags a read: SetIgnore, commands/overlap.go:203-223, only ever takes reads of at
    most two overlap sizes or reads within 10 % of the covered span - see the next test.)"""
    _overlap_against_fixture(_golden(case), slots=5, max_rounds=24)


def test_flagged_reads_at_full_scale_match_oracle_fixture():
    """The speculation / re-queue machinery at full scale: 400 000 reads of 1.5-4.5 kb (1.2 Gbase, served by the k-mer position
    index like config 2).  Every read of at most 2 000 bases that gets a consensus is flagged (commands/overlap.go:203-205), 779
    of them in the first 24 rounds, so executor slots see rounds rejected and planner lanes see their guesses fail; PAF and
    ignore flags against the oracle's fixture."""
    g = _golden("short_variable_k13_first_24_rounds")
    assert g["ignored_reads"] > 500
    st = _overlap_against_fixture(g, slots=5, max_rounds=g["max_rounds"])
    assert st["idx_rounds"] == 1


def test_config3_full_size_map_matches_oracle():
    """BASELINE config 3: 50 000 reads x 8 kb (10 % error) against a 4.6 Mb circular reference, k=11."""
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    G, N, L, e, seed = 4600000, 50000, 8000, 0.1, 3
    genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off = O.gen_reads(seed, G, N, L, e, False)
    want, werr = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=500, himem=False),
                           circular=True, k=11)
    got, gerr, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=500, himem=False),
                              circular=True, k=11)
    assert want.count("\n") > N // 2
    assert hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.encode()).hexdigest()
    assert got == want and gerr == werr


@pytest.mark.parametrize("shards", [1, 8])
def test_config5_one_gpu_share_of_the_reference(monkeypatch, shards):
    """BASELINE config 5 is a 3 Gb reference spread over 8 GPUs: 375 Mb of reference per GPU.  One such share at k = 13 -
    37 880 reference chunks, 2.9 M seeds: posting and seed-set bit matrices of 13.7 GB each - is indexed on the one GPU here
    and 15 kb reads (10 % error) are mapped against it; the oracle maps the first 300 of them against the same 375 Mb on the
    host (40 s).  Identical PAF - with the whole index in one context, and with config 5's own layout at this scale: the
    chunks dealt to eight contexts (DP_MAP_SHARDS=8; on an 8-GPU node DP_MAP_DEVICES=0,...,7 puts each on its own GPU),
    global set windows for the index query and the ratchets handed from shard to shard (include/downpore_hip.h,
    dp_index_set_global / dp_map_windows_shard)."""
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    if shards > 1:
        monkeypatch.setenv("DP_MAP_SHARDS", str(shards))
    from tools.synth import gen_reads_truth
    from tools.truth import map_truth
    G, N, L, e, seed, n_cpu = 375000000, 1000, 15000, 0.1, 5, 300
    genome = np.frombuffer(O.gen_genome(seed, G), dtype=np.uint8)
    goff = np.array([0, G], dtype=np.int64)
    bases, off, starts, strands = gen_reads_truth(seed, G, N, L, e, False)  # (the reads of O.gen_reads, with where they came from)
    got, gerr, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=500, himem=False),
                              circular=True, k=13)
    assert st["n_chunks"] > 37000 and st["n_seeds"] > 2000000
    # every one of the 1 000 reads held to the position the generator took it from (the oracle below covers the first 300 on the host)
    t = map_truth(got, off, starts, strands, G)
    print("config-5 share, %d shard(s): %s" % (shards, t))
    assert t["recall"] >= 0.99 and t["precision"] >= 0.99, t
    want, werr = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False),
                           O.ReadSet(bases[:off[n_cpu]], off[:n_cpu + 1], min_len=500, himem=False), circular=True, k=13)
    assert want.count("\n") >= n_cpu * 9 // 10
    assert got.startswith(want)


def test_config4_first_rounds_match_oracle_fixture():
    """BASELINE config 4's read set (1 M x 10 kb = 10 Gbase resident on ONE GPU, positions beyond 2^32, 80 GB k-mer index):
    the first rounds against the oracle's fixture.  (The 8-GPU form of config 4 is the driver's to run.)"""
    path = os.path.join(ROOT, "tests", "golden_full")
    names = [f[:-5] for f in os.listdir(path) if f.startswith("config4_first_") and f.endswith(".json")]
    assert names, "no config-4 fixture committed"
    g = _golden(sorted(names)[0])
    st = _overlap_against_fixture(g, slots=3, max_rounds=g["max_rounds"])
    assert st["idx_rounds"] == 1


def _stream_job(pipe, sample_every, truth):
    """Runs the job to its end, taking each step's PAF as bytes: line count, SHA-256 of the whole PAF, SHA-256 of its first
    `truth["head_lines"]` lines, and the ground-truth figures (tools/truth.py) summed over every `sample_every`-th step's text."""
    import ctypes as C
    from tools.truth import overlap_truth
    H = pipe.H
    sha, head = hashlib.sha256(), hashlib.sha256()
    head_left = truth["head_lines"]
    lines = rounds = steps = 0
    acc = {}
    while True:
        c = pipe.step()
        if c == 0:
            break
        n = C.c_int64(0)
        p = H.dph_overlap_round_paf(pipe.h, C.byref(n))
        text = C.string_at(p, n.value)
        sha.update(text)
        if head_left > 0:
            cut, pos = 0, -1
            while cut < head_left:
                pos = text.find(b"\n", pos + 1)
                if pos < 0:
                    break
                cut += 1
            head.update(text[:pos + 1] if pos >= 0 else text)
            head_left -= cut
        lines += text.count(b"\n")
        rounds += c
        if steps % sample_every == 0:
            t = overlap_truth(text, truth["off"], truth["starts"], truth["strands"], truth["k"], sample=20000)
            if t:
                w = t["lines_checked"]
                for key, v in t.items():
                    acc[key] = acc.get(key, 0) + (v * w if key.startswith("frac_") else v)
        steps += 1
    for key in list(acc):
        if key.startswith("frac_"):
            acc[key] /= acc["lines_checked"]
    return dict(lines=lines, rounds=rounds, sha=sha.hexdigest(), head_sha=head.hexdigest(), truth=acc)


def test_config4_whole_job_on_one_gpu():
    """BASELINE config 4's whole job on ONE GPU (1 M reads x 10 kb resident, 50 GB k-mer position index, ~6 000 rounds, ~38 M PAF
    lines): the first 6 rounds against the oracle's fixture as above; then ALL rounds through properties that need no second
    implementation - every sampled line joins two reads that really overlap on the synthetic genome, on the right relative strand,
    with both parts on the same stretch (tools/truth.py) - and the same job in a second executor layout (3 slots instead of 5,
    k-mer index rebuilt) must print the same PAF byte for byte (SHA-256 over ~2.3 GB of text) and flag the same reads."""
    from downpore_amd.overlap import OverlapPipeline, Reads
    from tools.synth import gen_reads_truth
    g = _golden("config4_first_6_rounds")
    gen = g["generator"]
    bases, off, starts, strands = gen_reads_truth(gen["seed"], gen["genome"], gen["reads"], gen["read_len"], gen["error"], gen["variable"])
    reads = Reads(bases, off, min_len=1000)
    del bases
    truth = dict(off=off, starts=starts, strands=strands, k=g["k"], head_lines=g["paf_lines"])
    results, ignores = [], []
    pipe = OverlapPipeline(reads, device=0, k=g["k"], slots=5, defer_init=True)  # (reads uploaded and packed once)
    for slots in (5, 3):
        pipe._params[7] = slots
        pipe.init()
        r = _stream_job(pipe, sample_every=16, truth=truth)
        st = pipe.stats()
        ignores.append(hashlib.sha256(reads.ignore().tobytes()).hexdigest())
        pipe.reset()  # (ends the job: slots, planner, value table and k-mer index released, ignore flags cleared)
        print("config 4, %d slots: %d rounds, %d lines, truth %s" % (slots, r["rounds"], r["lines"], r["truth"]))
        assert st["idx_rounds"] == 1
        assert r["head_sha"] == g["paf_sha256"], "the first %d lines are not the oracle's first 6 rounds" % g["paf_lines"]
        assert r["rounds"] > 5500 and r["lines"] > 30000000
        t = r["truth"]
        assert t["lines_checked"] > 150000  # (every 16th step of ~6 000 rounds, at most 20 000 lines each: how many depends on how the steps fall)
        assert t["strand_mismatches"] == 0
        assert t["reads_that_do_not_overlap_on_the_genome"] <= t["lines_checked"] // 100
        assert t["parts_without_a_shared_base"] <= t["lines_checked"] // 200
        assert t["frac_both_ends_within_30_bases"] >= 0.9 and t["frac_both_ends_within_k_bases"] >= 0.8
        results.append(r)
    assert results[0]["lines"] == results[1]["lines"] and results[0]["rounds"] == results[1]["rounds"]
    assert results[0]["sha"] == results[1]["sha"], "two executor layouts printed different PAF"
    assert ignores[0] == ignores[1]
    pipe.close()
