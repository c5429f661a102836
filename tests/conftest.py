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


# ---- which das arithmetic a test runs ---------------------------------------------------------------------------------------------
# The library default is the reference's arithmetic (BF_DAS_F64, das.cpp:16-24).  Modules about the das node ask for `das_impls`
# (pytestmark usefixtures): each of their tests then runs once per arithmetic -- the default and the BF_DAS_FUSED_F32 opt-in -- through
# conftest.Beamformer, which fills in das_impl wherever the test does not name one.  `das_f32` pins the opt-in (kernels that exist in
# fp32 only: shared look directions, the register-resident kernels of the other JACK periods).
# `precisions` does the same for bf_config.precision: BF_PRECISION_REFERENCE (the default: complex doubles between the transforms, fp64
# backward transform) and the BF_PRECISION_MIXED opt-in (mvdr / lcmv: z48 spectra; fp32 backward transform where a node can feed it).
_das_impl = {"value": None}
_precision = {"value": None}


def Beamformer(params, **kw):
    from beamform_amd import capi
    if _das_impl["value"] is not None and "das_impl" not in kw:
        kw["das_impl"] = _das_impl["value"]
    if _precision["value"] is not None and "precision" not in kw:
        kw["precision"] = _precision["value"]
    return capi.Beamformer(params, **kw)


@pytest.fixture(params=["reference", "mixed"])
def precisions(request):
    from beamform_amd import capi
    cs = getattr(request.node, "callspec", None)
    algo = cs.params.get("algo") if cs is not None else None
    if request.param == "mixed" and algo in ("gss", "mcra", "gsc"):
        pytest.skip("BF_PRECISION_MIXED changes nothing for this node")
    _precision["value"] = capi.BF_PRECISION_REFERENCE if request.param == "reference" else capi.BF_PRECISION_MIXED
    yield request.param
    _precision["value"] = None


@pytest.fixture(params=["f64", "f32"])
def das_impls(request):
    from beamform_amd import capi
    cs = getattr(request.node, "callspec", None)
    algo = cs.params.get("algo") if cs is not None else None
    if request.param == "f32" and isinstance(algo, str) and not algo.startswith("das"):
        pytest.skip("the fp32 opt-in only concerns das")
    _das_impl["value"] = capi.BF_DAS_F64 if request.param == "f64" else capi.BF_DAS_FUSED_F32
    yield request.param
    _das_impl["value"] = None


@pytest.fixture
def das_f32():
    from beamform_amd import capi
    _das_impl["value"] = capi.BF_DAS_FUSED_F32
    yield
    _das_impl["value"] = None


def Beamformer_f32(params, **kw):
    """The fp32 opt-in by name: for child processes (environment switches are read once per process) about kernels that exist in fp32 only."""
    from beamform_amd import capi
    kw.setdefault("das_impl", capi.BF_DAS_FUSED_F32)
    return capi.Beamformer(params, **kw)


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
