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
    import numpy as np
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
