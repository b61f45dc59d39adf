"""bench.py's launch contract, checked without a GPU: `--gpus N` must agree with WORLD_SIZE when a launcher set it, and a plain
`python bench.py --gpus N` (N > 1) must start N rank processes itself through torch.distributed.run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_must_match_world_size():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr


def test_plain_launch_builds_a_torchrun_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert bench.launch_ranks(8) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
