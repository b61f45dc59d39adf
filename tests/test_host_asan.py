"""The CPU tests of the host side - planner, window cache, finalCheck / consensus on the host, the FASTA / FASTQ reader - and of the
oracle's primitives once more against AddressSanitizer + UndefinedBehaviorSanitizer builds of libdownpore_host.so (`make host-asan`)
and of the checker (`make -C oracle asan`).  GPU sanitizers do not exist on this pool; the host code and the oracle are where an
out-of-bounds read would silently change a PAF line.  Runs in a child process (the sanitizer runtime has to be preloaded)."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    for pat in ("/usr/lib/gcc/x86_64-linux-gnu/*/lib%s.so" % name, "/usr/lib/x86_64-linux-gnu/lib%s.so*" % name):
        hits = sorted(glob.glob(pat))
        if hits:
            return hits[-1]
    return None


def test_host_and_oracle_cpu_tests_are_clean_under_asan_ubsan(tmp_path):
    asan, ubsan = _runtime("asan"), _runtime("ubsan")
    if asan is None:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "downpore_amd", "csrc"), "host-asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    log = str(tmp_path / "san")
    env = dict(os.environ, LD_PRELOAD=asan + (":" + ubsan if ubsan else ""),
               DPH_HOST_LIB=os.path.join(ROOT, "downpore_amd", "lib", "asan", "libdownpore_host.so"),
               DPO_LIB=os.path.join(ROOT, "oracle", "_build", "asan", "liboracle.so"),
               # (python itself leaks by design; an error must not be lost in a passing exit code: it is found in the log)
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=0:exitcode=0:log_path=" + log,
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=" + log)
    tests = ["tests/test_planner_epoch.py", "tests/test_host_coroutines.py::test_library_hook_switches_read_tasks_on_recycled_stacks", "tests/test_host_finalcheck.py", "tests/test_fasta_reader.py", "tests/test_oracle_known_answers.py",
             "tests/test_golden.py"]
    p = subprocess.run([sys.executable, "-m", "pytest"] + tests + ["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=2400)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    reports = [open(f).read() for f in glob.glob(log + ".*")]
    bad = [r for r in reports if ("AddressSanitizer" in r or "runtime error" in r) and ("libdownpore_host.so" in r or "liboracle.so" in r)]
    assert not bad, bad[0][:4000]
