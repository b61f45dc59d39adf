"""The mapper's read tasks are stackful coroutines (downpore_amd/csrc/host/host_coro.hpp).  The switch is hand-written for x86-64 and
falls back to <ucontext.h> elsewhere and under the sanitizers, which are told about every stack switch.  A `map` run needs a GPU; the
switch itself does not: the library's own hook and a stand-alone program built in every flavour of the header."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "downpore_amd", "csrc", "host")


def test_library_hook_switches_read_tasks_on_recycled_stacks():
    from downpore_amd.overlap import load_host
    H = load_host()
    H.dph_test_coroutines.restype = C.c_long
    H.dph_test_coroutines.argtypes = [C.c_int, C.c_int]
    assert H.dph_test_coroutines(0, 5) == 0
    assert H.dph_test_coroutines(1, 0) == 0
    assert H.dph_test_coroutines(100, 9) == 9 * 100 * 101 // 2


@pytest.mark.parametrize("name,flags,expect", [
    ("asm", ["-fcf-protection=none"], "asm"),
    ("ucontext", ["-DDPH_CORO_UCONTEXT"], "ucontext"),
    ("cet", ["-fcf-protection=full"], "ucontext"),
    ("asan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"], "ucontext"),
    ("tsan", ["-fsanitize=thread"], "ucontext"),
])
def test_every_flavour_of_the_switch(tmp_path, name, flags, expect):
    exe = str(tmp_path / ("coro_" + name))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-I" + HOST] + flags + [os.path.join(ROOT, "tests", "native", "coro_check.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and name in ("asan", "tsan") and "cannot find" in b.stderr:
        pytest.skip("no sanitizer runtime in this image")
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    if name == "tsan" and "unexpected memory mapping" in r.stderr:
        pytest.skip("this kernel's address-space layout is one the image's ThreadSanitizer runtime refuses to start under")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.startswith(expect), r.stdout
