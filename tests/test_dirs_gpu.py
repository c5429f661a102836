"""Look-direction batches (SURVEY 8(e) "look directions", 8(f) row 4): n_dirs beams from the SAME input in one call.

Every direction must equal an independent reference node steered to that angle (the oracle run once per angle)."""
import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("das_impls")]

TOL_SPECTRUM = 1e-5  # north_star: per-frame relative L2 on the complex spectrum
TOL_TIME = 1e-5


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _herm(Y):
    N = Y.shape[-1]
    idx = (-np.arange(N)) % N
    return 0.5 * (Y + np.conj(Y[..., idx]))


def run_dirs(p, x, thetas, n_streams=1, **kw):
    """-> (bf, y [S*D, F*512], Y [S*D, F, 1024]) through bf_process_batch_device."""
    from conftest import Beamformer
    torch = _torch()
    F = x.shape[-1] // 512
    D = len(thetas)
    bf = Beamformer(p, n_streams=n_streams, n_dirs=D, **kw)
    bf.set_thetas(thetas)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.empty((n_streams * D, F * 512), dtype=torch.float32, device="cuda")
    Yd = torch.empty((n_streams * D, F, 1024, 2), dtype=torch.float64, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr())
    torch.cuda.synchronize()
    rms = bf.stream_rms(yd.data_ptr(), F)
    return bf, yd.cpu().numpy(), Yd.cpu().numpy().view(np.complex128)[..., 0], rms


def oracle_dir(p, x, theta):
    import oracle
    q = dict(p, theta=theta)
    return oracle.OracleNode(q).process(x, want_spectrum=True)


def check_dir(y, Y, y_ref, Y_ref, hermitian):
    fin = np.isfinite(Y_ref).all(axis=1)
    assert (np.isfinite(Y).all(axis=1) == fin).all()
    if hermitian:  # das: the fp32 kernels dump the Hermitian part of y_fft (what reaches Re(ifft)), the fp64 pipeline all of it
        Y, Yr = _herm(Y), _herm(Y_ref)
    else:
        Yr = Y_ref
    worst = max(rel_l2(Y[t], Yr[t]) for t in range(len(fin)) if fin[t])
    assert worst < TOL_SPECTRUM, worst
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y) == ok).all()
    assert rel_l2(y[ok], y_ref[ok]) < TOL_TIME


@pytest.mark.parametrize("M,F,thetas", [(8, 40, [-90.0, -30.0, 0.0, 20.0, 75.0]), (3, 21, [10.0, -170.0]),
                                        (16, 18, [0.0, 45.0, 90.0])])
def test_das_fused_look_directions(M, F, thetas):
    p = make_params("das", n_mics=M, theta=0.0)
    x = make_scene(M, F, seed=1000 + M)
    _, y, Y, rms = run_dirs(p, x, thetas)
    for d, th in enumerate(thetas):
        y_ref, Y_ref = oracle_dir(p, x, th)
        check_dir(y[d], Y[d], y_ref, Y_ref, hermitian=True)
        assert abs(rms[0, d] - np.sqrt(np.mean(y[d].astype(np.float64) ** 2))) < 1e-9  # energy2theta.py:23-27
    # the scene's target sits at 20 degrees: that beam must carry the most energy of the five
    if M == 8:
        assert int(np.argmax(rms[0])) == thetas.index(20.0)


def test_das_fused_streams_times_directions_and_long_batch():
    """2 input streams x 3 directions; a batch long enough for several frame runs per output stream."""
    M, F, S, thetas = 4, 600, 2, [-45.0, 0.0, 60.0]
    p = make_params("das", n_mics=M)
    xs = np.stack([make_scene(M, F, seed=50 + s) for s in range(S)])
    _, y, Y, _ = run_dirs(p, xs, thetas, n_streams=S)
    for s in range(S):
        for d, th in enumerate(thetas):
            y_ref, Y_ref = oracle_dir(p, xs[s], th)
            check_dir(y[s * 3 + d], Y[s * 3 + d], y_ref, Y_ref, hermitian=True)


@pytest.mark.parametrize("algo,M,interf,F", [("phase", 8, (), 24), ("mvdr", 8, (), 40), ("mvdr", 16, (), 24),
                                             ("lcmv", 8, (-60.0, 90.0), 36), ("das", 8, (), 16)])
def test_bin_pipeline_look_directions(algo, M, interf, F):
    """fp64 pipeline: one STFT (and one covariance history) shared by all directions."""
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32
    thetas = [20.0, -35.0, 110.0]
    p = make_params(algo, n_mics=M, interf=interf)
    x = make_scene(M, F, seed=1100 + M)
    _, y, Y, _ = run_dirs(p, x, thetas, das_impl=BF_DAS_F64 if algo == "das" else BF_DAS_FUSED_F32)
    for d, th in enumerate(thetas):
        y_ref, Y_ref = oracle_dir(p, x, th)
        check_dir(y[d], Y[d], y_ref, Y_ref, hermitian=False)


@pytest.mark.parametrize("algo,M,interf,F", [("phasempf", 8, (), 60), ("gss", 8, (-60.0, 90.0), 40), ("gss", 4, (), 30)])
def test_recursive_nodes_keep_their_state_per_beam(algo, M, interf, F):
    """gss and phasempf recurse on their own output: every beam carries its own demixing matrices / noise estimates."""
    thetas = [20.0, -35.0, 110.0]
    p = make_params(algo, n_mics=M, interf=interf)
    x = make_scene(M, F, seed=1300 + M)
    _, y, Y, _ = run_dirs(p, x, thetas)
    for d, th in enumerate(thetas):
        y_ref, Y_ref = oracle_dir(p, x, th)
        check_dir(y[d], Y[d], y_ref, Y_ref, hermitian=False)


@pytest.mark.parametrize("algo", ["das", "mvdr", "gss"])
def test_directions_retarget_between_batches(algo):
    """bf_set_theta_dir between batches == set_theta on that direction's reference node; the others are untouched."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 8, 30
    p = make_params(algo, n_mics=M, interf=(-60.0,) if algo == "gss" else ())
    x = make_scene(M, F, seed=77)
    thetas = [0.0, 50.0]
    nodes = [oracle.OracleNode(dict(p, theta=t)) for t in thetas]
    bf = Beamformer(p, n_dirs=2)
    bf.set_thetas(thetas)
    cuts = [0, 7, 19, 30]
    ys, refs = [], [[], []]
    for n, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if n == 1:
            bf.set_theta_dir(1, -80.0)
            nodes[1].set_theta(-80.0)
        seg = np.ascontiguousarray(x[:, a * 512:b * 512])
        ys.append(bf.process(seg))
        for d in range(2):
            refs[d].append(nodes[d].process(seg)[0])
    y = np.concatenate(ys, axis=1)
    for d in range(2):
        r = np.concatenate(refs[d])
        ok = np.isfinite(r)
        assert (np.isfinite(y[d]) == ok).all()
        assert rel_l2(y[d][ok], r[ok]) < TOL_TIME


def test_hop_by_hop_with_directions_and_checkpoint():
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 4, 9
    p = make_params("das", n_mics=M)
    x = make_scene(M, F, seed=5)
    thetas = [15.0, -120.0]
    bf = Beamformer(p, n_dirs=2)
    bf.set_thetas(thetas)
    hops = []
    for t in range(F):
        if t == 4:  # restore into a fresh handle mid-stream
            blob = bf.get_state()
            bf = Beamformer(p, n_dirs=2)
            bf.set_thetas(thetas)
            bf.set_state(blob)
        hops.append(bf.process_hop(x[:, t * 512:(t + 1) * 512]))
    y = np.concatenate(hops, axis=1)
    for d, th in enumerate(thetas):
        assert rel_l2(y[d], oracle_dir(p, x, th)[0]) < TOL_TIME


def test_directions_rejected_where_meaningless():
    from beamform_amd.capi import BfError
    from conftest import Beamformer
    _torch()
    for algo in ("mcra", "gsc"):
        with pytest.raises(BfError) as e:
            Beamformer(make_params(algo, n_mics=4), n_dirs=2)
        assert e.value.code == -38  # BF_ENOSYS
