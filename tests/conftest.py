import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emul_lib():
    """CPU emulation of the half-wavefront kernels (tests/host_emul/emul.cpp), built with g++."""
    import ctypes
    so = os.path.join(ROOT, "tests", "host_emul", "libemul.so")
    src = os.path.join(ROOT, "tests", "host_emul", "emul.cpp")
    hdrs = [os.path.join(ROOT, "beamform_amd", "csrc", f) for f in ("fft32.hpp", "fft1024.hpp", "fft1024_w64.hpp", "geometry.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


def rel_l2(a, b):
    """||a - b|| / ||b||.  Both operands are scaled by the reference's largest magnitude first, so spectra of order 1e200 (the
    rounding noise of a numerically singular constraint system) do not overflow the norm into inf / inf; a NaN result -- any
    non-finite operand -- fails here instead of slipping through a max() (Python's max drops NaN unless it comes first)."""
    import numpy as np
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    assert np.isfinite(a).all() and np.isfinite(b).all(), "rel_l2: non-finite operand (mask those frames out explicitly)"
    s = float(np.abs(b).max()) if b.size else 0.0
    if s > 0.0:
        a, b = a / s, b / s
    r = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
    assert not np.isnan(r), "rel_l2 is NaN"
    return r
