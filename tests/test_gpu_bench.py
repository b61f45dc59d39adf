"""bench.py as the driver runs it, on the one GPU of the test box: the single-GPU line and - with the two test hooks that let two
ranks share a GPU (DP_BENCH_SAME_DEVICE=1, exchange over gloo) - both multi-GPU decompositions as real processes under
torch.distributed.run.  Every run hashes a whole config-2 job against the oracle fixture (parity.paf_sha256_matches_oracle_fixture)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    return json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])


def test_bench_single_gpu_contract():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-rounds", "0", "--scan-leg-rounds", "0",
                        "--dense-leg-rounds", "0"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["metric"].startswith("overlaps/sec") and d["unit"] == "overlaps/s" and d["vs_baseline"] is None
    assert d["parity"]["paf_sha256_matches_oracle_fixture"] is True and d["parity"]["jobs_with_equal_line_count"] == 2
    assert d["value"] > 1e6 and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert "workload" in d["config"]


@pytest.mark.parametrize("mode", ["round", "scan-shard"])
def test_bench_two_ranks_sharing_the_gpu(mode):
    env = dict(os.environ, DP_BENCH_SAME_DEVICE="1", DP_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-rounds", "0", "--mode", mode,
                        "--slots", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["parity"]["paf_sha256_matches_oracle_fixture"] is True


def test_bench_two_ranks_default_reports_both_layouts():
    """N > 1 without --mode: the headline is the layout that scales while a GPU holds the reads (query batches dealt to the ranks,
    results all-gathered; every rank plans its own rounds only), the reads-partitioned layout (survivors all-gathered) follows as
    `alt_mode`; each is held to the fixture on its own."""
    env = dict(os.environ, DP_BENCH_SAME_DEVICE="1", DP_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29518", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-rounds", "0", "--slots", "4"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and "query batches dealt to 2 GPUs" in d["config"]["parallelism"]
    assert d["parity"]["paf_sha256_matches_oracle_fixture"] is True
    assert d["alt_mode"]["mode"] == "scan-shard" and d["alt_mode"]["paf_sha256_matches_oracle_fixture"] is True and d["alt_mode"]["value"] > 0
    assert len(d["per_rank"]) == 2 and all(r["per_job"]["plans_computed"] < 0.75 * d["config"]["rounds_per_step"] for r in d["per_rank"])


def test_bench_plain_launch_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent starts two rank processes itself (fresh children of a
    process that has not touched the GPU), relays rank 0's line and the exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DP_BENCH_SAME_DEVICE="1", DP_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-rounds", "0", "--mode", "round",
                        "--slots", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["parity"]["paf_sha256_matches_oracle_fixture"] is True


def test_bench_two_gpus_over_rccl():
    """Two real ranks, one GPU each, exchanging over RCCL inside the library (dp_comm_init / dp_allgather_survivors /
    dp_allgather_blobs): what the driver's multi-GPU run does.  Needs a second GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--cpu-rounds", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["parity"]["paf_sha256_matches_oracle_fixture"] is True
    assert d["alt_mode"]["paf_sha256_matches_oracle_fixture"] is True
