"""CPU tests of the boundary: the C-ABI library loads and exports every symbol that
include/praline_dp.h declares; compute entry points fail loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "praline_dp.h")
LIB = os.path.join(ROOT, "praline_amd", "libpraline_dp.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(praline_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(LIB)
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), "missing export: " + name


def test_no_silent_cpu_fallback():
    from praline_amd import native
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    p = np.zeros((4, 27), dtype=np.float32)
    p[:, 0] = 1
    with pytest.raises(native.NativeError):
        native.Arena([p, p], np.eye(27, dtype=np.float32))
    m = np.zeros((4, 4), dtype=np.float32)
    with pytest.raises(native.NativeError):
        native.cext_build_scores([p], [p], None, None, [np.eye(27, dtype=np.float32)], m)
