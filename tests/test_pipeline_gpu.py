"""GPU parity of the fp64 bin pipeline (STFT -> per-bin kernel -> ISTFT) against the CPU oracle."""
import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_SPECTRUM = 1e-5   # north_star: 1e-5 relative on the complex spectrum (per-frame relative L2)
TOL_TIME = 1e-5


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def run_gpu(p, x, n_streams=1, **kw):
    """-> (y [F*512] float32, Y [F,1024] complex128) through bf_process_batch_device."""
    from beamform_amd.capi import Beamformer
    torch = _torch()
    F = x.shape[-1] // 512
    bf = Beamformer(p, n_streams=n_streams, **kw)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.empty((n_streams, F * 512), dtype=torch.float32, device="cuda")
    Yd = torch.empty((n_streams, F, 1024, 2), dtype=torch.float64, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr())
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    Y = Yd.cpu().numpy().view(np.complex128)[..., 0]
    return (y[0], Y[0]) if n_streams == 1 else (y, Y)


def check(y, Y, y_ref, Y_ref, skip=0):
    F = Y_ref.shape[0]
    fin = np.isfinite(Y_ref).all(axis=1)
    # frames the reference itself turns into NaN/Inf (mvdr/lcmv frame 0: inverse of the zero matrix) must be
    # non-finite on the GPU as well
    assert (np.isfinite(Y).all(axis=1) == fin).all()
    worst = max(rel_l2(Y[t], Y_ref[t]) for t in range(skip, F) if fin[t])
    assert worst < TOL_SPECTRUM, worst
    ok = np.isfinite(y_ref) & np.isfinite(y)
    assert (np.isfinite(y_ref) == np.isfinite(y)).all()
    assert rel_l2(y[ok], y_ref[ok]) < TOL_TIME


@pytest.mark.parametrize("M,theta,F", [(8, 20.0, 32), (4, -35.0, 17), (3, 60.0, 9), (16, 90.0, 12)])
def test_das_bins_f64_full_spectrum(M, theta, F):
    """The fp64 path reproduces the reference's FULL 1024-bin y_fft, including the non-conjugate Q1 bins."""
    import oracle
    from beamform_amd.capi import BF_DAS_BINS_F64
    p = make_params("das", n_mics=M, theta=theta)
    x = make_scene(M, F, seed=300 + M)
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    y, Y = run_gpu(p, x, das_impl=BF_DAS_BINS_F64)
    check(y, Y, y_ref, Y_ref)
    assert np.abs(Y - Y_ref).max() < 1e-9 * np.abs(Y_ref).max()


@pytest.mark.parametrize("M,theta,F", [(8, 20.0, 40), (4, 0.0, 21), (2, 45.0, 8)])
def test_phase_matches_oracle(M, theta, F):
    import oracle
    p = make_params("phase", n_mics=M, theta=theta)
    x = make_scene(M, F, seed=400 + M)
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    y, Y = run_gpu(p, x)
    check(y, Y, y_ref, Y_ref)
