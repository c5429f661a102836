"""GPU parity of the das node against the CPU oracle (through the C ABI), once per arithmetic: double like the reference (the default:
das_f64_pair_kernel / das_f64_w64_kernel / the fp64 bin pipeline with a spectrum dump) and the fused fp32 opt-in (conftest.das_impls)."""
import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("das_impls")]

# north_star tolerance: 1e-5 relative on the complex spectrum (per-frame relative L2)
TOL_SPECTRUM = 1e-5
# time-domain output: fp32 storage on both sides; same budget
TOL_TIME = 1e-5


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _herm(Y):
    """Hermitian part (Y[k] + conj(Y[N-k]))/2 -- the part of y_fft that reaches Re(ifft)."""
    N = Y.shape[-1]
    idx = (-np.arange(N)) % N
    return 0.5 * (Y + np.conj(Y[..., idx]))


@pytest.mark.parametrize("M,theta,F", [(8, 0.0, 40), (8, 20.0, 64), (4, -35.0, 33), (3, 60.0, 17), (16, 90.0, 24), (1, 0.0, 5)])
def test_das_fused_matches_oracle(M, theta, F):
    import oracle
    from conftest import Beamformer
    torch = _torch()
    mics = None if M <= 16 else None
    p = make_params("das", n_mics=M, theta=theta)
    x = make_scene(M, F, seed=100 + M, mics=p["mics"])
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)

    bf = Beamformer(p)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    Yd = torch.empty((F, 1024, 2), dtype=torch.float64, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr())
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    Yh = _herm(Yd.cpu().numpy().view(np.complex128)[..., 0])   # (the fp32 kernel dumps the Hermitian part, the fp64 pipeline the full y_fft)

    assert rel_l2(y, y_ref) < TOL_TIME
    Yh_ref = _herm(Y_ref)
    per_frame = [rel_l2(Yh[t], Yh_ref[t]) for t in range(F)]
    assert max(per_frame) < TOL_SPECTRUM, max(per_frame)


def test_das_streaming_equals_batch_and_oracle():
    """bf_process_hop one callback at a time == one batch == oracle; state carries across calls."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 4, 12
    p = make_params("das", n_mics=M, theta=15.0)
    x = make_scene(M, F, seed=5)
    y_ref, _ = oracle.OracleNode(p).process(x)
    bf = Beamformer(p)
    y_hop = np.concatenate([bf.process_hop(x[:, t * 512:(t + 1) * 512]) for t in range(F)])
    assert rel_l2(y_hop, y_ref) < TOL_TIME
    bf2 = Beamformer(p)
    y_a = bf2.process(x[:, : 5 * 512])
    y_b = bf2.process(np.ascontiguousarray(x[:, 5 * 512:]))
    assert rel_l2(np.concatenate([y_a, y_b]), y_ref) < TOL_TIME
    # the first hop out is only the first half of frame 0 (latency of one hop, util.h:301-302)
    assert np.abs(y_hop[:512] - y_ref[:512]).max() < 1e-6


def test_das_set_theta_takes_effect_next_batch():
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 8, 10
    p = make_params("das", n_mics=M, theta=0.0)
    x = make_scene(M, 2 * F, seed=9)
    node = oracle.OracleNode(p)
    y1, _ = node.process(np.ascontiguousarray(x[:, : F * 512]))
    node.set_theta(-50.0)
    y2, _ = node.process(np.ascontiguousarray(x[:, F * 512:]))
    bf = Beamformer(p)
    z1 = bf.process(np.ascontiguousarray(x[:, : F * 512]))
    bf.set_theta(-50.0)
    z2 = bf.process(np.ascontiguousarray(x[:, F * 512:]))
    assert rel_l2(z1, y1) < TOL_TIME and rel_l2(z2, y2) < TOL_TIME
    assert np.abs(bf.weights() - node.weights()).max() < 1e-14


@pytest.mark.parametrize("M,F,S", [(8, 9, 3), (4, 40, 2), (8, 300, 1), (6, 21, 2), (3, 10, 1)])
def test_das_interleaved_layout_and_streams(M, F, S):
    """[sample][mic] input: 4 and 8 microphones take the 16-byte-load kernel (two pairs per access), the rest the generic one;
    F = 300 crosses run boundaries (atomic first hops) and ends inside a 16-frame iteration."""
    import oracle
    from beamform_amd.capi import BF_INTERLEAVED
    from conftest import Beamformer
    _torch()
    p = make_params("das", n_mics=M, theta=33.0)
    xs = [make_scene(M, F, seed=20 + s) for s in range(S)]
    refs = [oracle.OracleNode(p).process(x)[0] for x in xs]
    planar = np.stack(xs)                          # [S, M, T]
    y = np.atleast_2d(Beamformer(p, n_streams=S).process(planar))
    inter = np.ascontiguousarray(planar.transpose(0, 2, 1))  # [S, T, M]
    yi = np.atleast_2d(Beamformer(p, n_streams=S, layout=BF_INTERLEAVED).process(inter))
    for s in range(S):
        assert rel_l2(y[s], refs[s]) < TOL_TIME
        assert rel_l2(yi[s], refs[s]) < TOL_TIME


def test_das_checkpoint_roundtrip():
    from conftest import Beamformer
    _torch()
    M, F = 8, 8
    p = make_params("das", n_mics=M, theta=10.0)
    x = make_scene(M, 2 * F, seed=2)
    a = Beamformer(p)
    ya1 = a.process(np.ascontiguousarray(x[:, : F * 512]))
    blob = a.get_state()
    ya2 = a.process(np.ascontiguousarray(x[:, F * 512:]))
    b = Beamformer(p)
    b.set_state(blob)
    yb2 = b.process(np.ascontiguousarray(x[:, F * 512:]))
    assert np.array_equal(ya2, yb2)


def test_das_large_batch_properties():
    """BASELINE size (64k frames, 8 mics): size-independent checks instead of the oracle.
    (a) WOLA identity: identical signal on all mics at zero steering delay -> output = input delayed by one hop
        (util.h:301-302 + sqrt-Hann^2 COLA; the only known-answer property the reference offers, jack_ref.cpp);
    (b) linearity: DAS(a*x1 + x2) == a*DAS(x1) + DAS(x2);
    (c) chunk independence: a sub-batch cut out of the middle reproduces the same samples."""
    from conftest import Beamformer
    torch = _torch()
    M, F = 8, 65536
    co = [(0.0, 0.0)] * M                     # co-located mics: every delay is 0
    p = make_params("das", n_mics=M, mics=co)
    g = torch.Generator(device="cuda").manual_seed(1)
    s = (torch.rand(F * 512, device="cuda", generator=g) - 0.5)
    x = s.repeat(M, 1).contiguous()
    y = torch.empty(F * 512, device="cuda")
    Beamformer(p).process_device(x.data_ptr(), F, y.data_ptr())
    torch.cuda.synchronize()
    err = (y[512:] - s[:-512]).abs().max().item()
    assert err < 2e-6, err
    assert y[:512].abs().max().item() <= s[:512].abs().max().item() + 1e-6

    p2 = make_params("das", n_mics=M, theta=25.0)
    x1 = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
    x2 = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
    outs = []
    for xx in (x1, x2, (0.5 * x1 + x2)):
        o = torch.empty(F * 512, device="cuda")
        Beamformer(p2).process_device(xx.contiguous().data_ptr(), F, o.data_ptr())
        torch.cuda.synchronize()
        outs.append(o)
    lin = (outs[2] - (0.5 * outs[0] + outs[1])).norm() / outs[2].norm()
    assert lin.item() < 1e-6, lin.item()

    t0, n = 30000, 700
    sub = x1[:, (t0 - 1) * 512:(t0 + n) * 512].contiguous()      # one extra hop of history in front
    o = torch.empty((n + 1) * 512, device="cuda")
    Beamformer(p2).process_device(sub.data_ptr(), n + 1, o.data_ptr())
    torch.cuda.synchronize()
    # from the 2nd hop on, history matches the big run exactly
    d = (o[1024:] - outs[0][(t0 + 1) * 512:(t0 + n) * 512]).abs().max().item()
    assert d < 1e-6, d


def test_das_large_batch_random_oracle_windows():
    """BASELINE size against the ORACLE: a frame's output hop depends on three input hops only (util.h:217-242,301-302), so the
    oracle fed hops [t0 - 2, t0 + n) reproduces hops t0 .. t0 + n - 1 of the 65 536-frame batch exactly as the reference would
    compute them in the middle of the stream.  Windows at random offsets, at run boundaries of the kernel's frame walk
    (multiples of 256 and of 16) and at both ends of the batch."""
    import oracle
    from conftest import Beamformer
    torch = _torch()
    M, F, n = 8, 65536, 24
    p = make_params("das", n_mics=M, theta=-40.0)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
    y = torch.empty(F * 512, device="cuda")
    Beamformer(p).process_device(x.data_ptr(), F, y.data_ptr())
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    starts = [0, 2, 254, 256 * 100 - 3, 256 * 255 + 9, 16 * 1234 - 1, F - n] + [int(v) for v in rng.integers(2, F - n, 6)]
    for t0 in starts:
        a = max(t0 - 2, 0)
        seg = x[:, a * 512:(t0 + n) * 512].cpu().numpy()
        y_ref, _ = oracle.OracleNode(p).process(np.ascontiguousarray(seg))
        ref = y_ref[(t0 - a) * 512:]
        got = y[t0 * 512:(t0 + n) * 512].cpu().numpy()
        assert rel_l2(got, ref) < TOL_TIME, t0


@pytest.mark.parametrize("M", [8, 4])
def test_das_interleaved_equals_planar_at_the_baseline_size(M):
    """65 536 frames: the 16-byte-load kernel for [sample][mic] input against the planar kernel on the transposed data
    (same transform; the two kernels round the window multiply differently, hence a tolerance of a few float ulps)."""
    torch = _torch()
    from beamform_amd.capi import BF_INTERLEAVED
    from conftest import Beamformer
    F = 65536
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
    xi = x.t().contiguous()
    y, yi = torch.empty(F * 512, device="cuda"), torch.empty(F * 512, device="cuda")
    p = make_params("das", n_mics=M, theta=-15.0)
    s = torch.cuda.current_stream().cuda_stream
    Beamformer(p).process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
    Beamformer(p, layout=BF_INTERLEAVED).process_device(xi.data_ptr(), F, yi.data_ptr(), 0, s)
    torch.cuda.synchronize()
    assert float((y - yi).abs().max()) <= 4e-7 * float(y.abs().max())
