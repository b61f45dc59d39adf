"""Golden fixtures (tests/golden/*.json, written by tools/make_golden.py from the oracle on seeded synthetic inputs).
CPU: the oracle must still reproduce them (no silent drift of the checker).  GPU: the product pipeline must print the
same PAF round by round and flag the same reads."""
import glob
import hashlib
import json
import os

import numpy as np
import pytest

from tests import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(p for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json")) if not os.path.basename(p).startswith("map_"))
MAP_FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "map_*.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load(path):
    fx = json.load(open(path))
    g = fx["generator"]
    bases, off = O.gen_reads(g["seed"], g["genome"], g["reads"], g["read_len"], g["error"], g["variable"])
    b = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
    assert _sha(b) == fx["input_sha256"], "synthetic generator changed"
    return fx, bases, off


def test_fixtures_exist():
    assert len(FIXTURES) >= 6 and len(MAP_FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    fx, bases, off = _load(path)
    rs = O.ReadSet(bases, off, min_len=1000)
    run = O.OverlapRun(rs, k=fx["k"], max_rounds=fx["max_rounds"], traces=True, **fx["kwargs"])
    assert run.rounds == fx["n_rounds"]
    for r, want in enumerate(fx["rounds"]):
        assert _sha(run.trace(r, "seedKmers").astype(np.int64)) == want["seed_kmers_sha256"], (r, "seeds")
        isg, io = run.trace(r, "indexedSegments")
        assert _sha(isg.astype(np.int64)) == want["indexed_segments_sha256"], (r, "indexed segments")
        cd, _ = run.trace(r, "candidates")
        assert _sha(cd.astype(np.int64)) == want["candidates_sha256"], (r, "candidates")
        ma, mo = run.trace(r, "matchA")
        mb, _ = run.trace(r, "matchB")
        assert len(mo) - 1 == want["n_matches"]
        assert _sha(ma.astype(np.int64)) == want["match_a_sha256"] and _sha(mb.astype(np.int64)) == want["match_b_sha256"]
        assert hashlib.sha256(run.trace_paf(r).encode()).hexdigest() == want["paf_sha256"], (r, "paf")
        assert [int(x) for x in run.trace(r, "newlyIgnored")] == want["newly_ignored"]
    assert hashlib.sha256(run.paf.encode()).hexdigest() == fx["paf_sha256"]
    assert run.paf.split("\n")[:12] == fx["paf_head"]
    assert int(rs.ignore().sum()) == fx["ignored_reads"]


@pytest.fixture(params=["host-consensus", "device-consensus"])
def consensus_mode(request):
    """The optional device-side consensus alignment (dp_consensus_align, DP_DEVICE_CONSENSUS=1) must not change a byte."""
    if request.param == "device-consensus":
        os.environ["DP_DEVICE_CONSENSUS"] = "1"
    yield request.param
    os.environ.pop("DP_DEVICE_CONSENSUS", None)


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 3])
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-5] for p in FIXTURES])
def test_gpu_pipeline_matches_golden(path, slots, consensus_mode):
    from downpore_amd.overlap import OverlapPipeline, Reads
    fx, bases, off = _load(path)
    reads = Reads(bases, off, min_len=1000)
    pipe = OverlapPipeline(reads, k=fx["k"], slots=slots, **fx["kwargs"])
    rounds = 0
    while fx["max_rounds"] < 0 or rounds < fx["max_rounds"]:
        c = pipe.step()
        if c == 0:
            break
        rounds += c
    paf = pipe.all_paf()
    if fx["max_rounds"] >= 0 and rounds > fx["max_rounds"]:  # a step may commit several rounds: compare the prefix
        paf = "".join(paf.splitlines(True)[:fx["paf_lines"]])
    else:
        assert rounds == fx["n_rounds"]
        assert int(reads.ignore().sum()) == fx["ignored_reads"]
    assert paf.count("\n") == fx["paf_lines"]
    assert hashlib.sha256(paf.encode()).hexdigest() == fx["paf_sha256"]
    pipe.close()


def _map_inputs(fx):
    g = fx["generator"]
    genome = np.frombuffer(O.gen_genome(g["seed"], g["genome"]), dtype=np.uint8)
    goff = np.array([0, g["genome"]], dtype=np.int64)
    bases, off = O.gen_reads(g["seed"], g["genome"], g["reads"], g["read_len"], g["error"], g["variable"])
    return genome, goff, bases, off


@pytest.mark.parametrize("path", MAP_FIXTURES, ids=[os.path.basename(p)[:-5] for p in MAP_FIXTURES])
def test_oracle_map_reproduces_golden(path):
    fx = json.load(open(path))
    genome, goff, bases, off = _map_inputs(fx)
    paf, err = O.map_run(O.ReadSet(genome, goff, min_len=0, himem=False), O.ReadSet(bases, off, min_len=500, himem=False),
                         circular=fx["circular"], k=fx["k"])
    assert paf.count("\n") == fx["paf_lines"] and hashlib.sha256(paf.encode()).hexdigest() == fx["paf_sha256"]
    assert err == fx["stderr"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", MAP_FIXTURES, ids=[os.path.basename(p)[:-5] for p in MAP_FIXTURES])
def test_gpu_map_matches_golden(path):
    from downpore_amd.mapping import map_reads
    from downpore_amd.overlap import Reads
    fx = json.load(open(path))
    genome, goff, bases, off = _map_inputs(fx)
    paf, err, st = map_reads(Reads(genome, goff, min_len=0, himem=False), Reads(bases, off, min_len=500, himem=False),
                             circular=fx["circular"], k=fx["k"])
    assert paf.count("\n") == fx["paf_lines"] and hashlib.sha256(paf.encode()).hexdigest() == fx["paf_sha256"]
    assert err == fx["stderr"]
