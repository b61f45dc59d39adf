"""The planner tests once more against the ThreadSanitizer build of the host library (`make host-tsan`): planner lanes, the window
cache's producer and the committer share the read set's flags and the plan chain - a data race there is a defect even when the
epoch validation happens to catch its effect.  Runs in a child process (the sanitizer runtime has to be preloaded)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libtsan():
    for pat in ("/usr/lib/gcc/x86_64-linux-gnu/*/libtsan.so", "/usr/lib/x86_64-linux-gnu/libtsan.so*"):
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[-1]
    return None


def test_planner_tests_are_race_free_under_tsan(tmp_path):
    tsan = _libtsan()
    if tsan is None:
        pytest.skip("no libtsan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "downpore_amd", "csrc"), "host-tsan"])
    lib = os.path.join(ROOT, "downpore_amd", "lib", "tsan", "libdownpore_host.so")
    log = str(tmp_path / "tsan")
    env = dict(os.environ, LD_PRELOAD=tsan, DPH_HOST_LIB=lib,
               TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=0 log_path=" + log)
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_planner_epoch.py", "tests/test_host_coroutines.py::test_library_hook_switches_read_tasks_on_recycled_stacks", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    reports = [open(f).read() for f in glob.glob(log + ".*")]
    races = [r for r in reports if "ThreadSanitizer: data race" in r and "libdownpore_host.so" in r]
    assert not races, races[0][:3000]
