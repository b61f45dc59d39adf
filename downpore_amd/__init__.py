"""downpore_amd — MI355X-native seed-index + seed-chaining overlap/map path (libdownpore_hip.so front end).

The package is a thin ctypes binding over the C ABI declared in include/downpore_hip.h plus the host-side
driver (C++: downpore_amd/csrc/host, built as bin/downpore).  There is no CPU fallback: loading fails loudly when
the HIP library has not been built, and creating a context fails when no GPU is present.
"""
from .hip import Context, DpError, lib_path, load_library  # noqa: F401
