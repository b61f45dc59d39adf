"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol the header declares;
context creation fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest


def _header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "downpore_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import downpore_amd.hip as h
    L = h.load_library()
    syms = _header_symbols()
    assert syms, "no symbols parsed from include/downpore_hip.h"
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(h.SYMBOLS) == syms
    assert b"gfx950" in L.dp_version()


def test_library_exports_nothing_else():
    """The dynamic symbol table holds the C ABI only (version script + -fvisibility=hidden): no kernel launch stubs, no
    C++ template instantiations, no internal helpers."""
    import subprocess
    import downpore_amd.hip as h
    out = subprocess.check_output(["nm", "-D", "--defined-only", h.lib_path()], text=True)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == _header_symbols()


def _host_header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "downpore_host.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dph_[a-z_0-9]+)\s*\(", txt)))


def test_host_library_exports_exactly_its_header():
    """libdownpore_host.so - the pipeline the bench measures - has a published boundary of its own (include/downpore_host.h):
    every declared entry point is exported, and nothing else is (version script: the C++ mirror of the Go layers stays local)."""
    import subprocess
    from downpore_amd.overlap import host_lib_path, load_host
    H = load_host()
    syms = _host_header_symbols()
    assert len(syms) > 50
    for s in syms:
        assert hasattr(H, s), s
    out = subprocess.check_output(["nm", "-D", "--defined-only", host_lib_path()], text=True)
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert exported == syms


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import downpore_amd
    with pytest.raises(downpore_amd.DpError):
        downpore_amd.Context(0)
