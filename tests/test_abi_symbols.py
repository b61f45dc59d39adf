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


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import downpore_amd
    with pytest.raises(downpore_amd.DpError):
        downpore_amd.Context(0)
