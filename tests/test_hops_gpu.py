"""Boundary generality on the GPU: every power-of-two JACK period from 64 to 4096 frames (rosjack.cpp:131-134 takes what the
server reports; fft_win = 2 * period, util.h:261),
more than three interferers (lcmv.cpp:258-309 appends without a cap; beamform_config.yaml:43-57 lists 15) and more than
16 microphones -- every case against the oracle through the C ABI."""
import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("das_impls")]

TOL = 1e-5  # north_star tolerance: relative L2, per frame on the complex spectrum and on the time signal


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def run(p, x, n_streams=1):
    torch = _torch()
    from conftest import Beamformer
    H = p["hop"]
    F = x.shape[-1] // H
    bf = Beamformer(p, n_streams=n_streams)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    yd = torch.empty((n_streams, F * H), dtype=torch.float32, device="cuda")
    Yd = torch.empty((n_streams, F, 2 * H, 2), dtype=torch.float64, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr())
    torch.cuda.synchronize()
    return bf, yd.cpu().numpy(), Yd.cpu().numpy().view(np.complex128)[..., 0]


def check(y, Y, y_ref, Y_ref, skip=0, hop=512):
    """skip: leading frames left out (covariance window still filling: with as many constraints as microphones the
    rank-deficient R of the first P frames makes inverse() / Cholesky return different non-finite patterns)."""
    y, Y, y_ref, Y_ref = y[skip * hop:], Y[skip:], y_ref[skip * hop:], Y_ref[skip:]
    fin = np.isfinite(Y_ref).all(axis=1)
    assert (np.isfinite(Y).all(axis=1) == fin).all()
    worst = max(rel_l2(Y[t], Y_ref[t]) for t in range(len(fin)) if fin[t] and np.abs(Y_ref[t]).max() > 0)
    assert worst < TOL, worst
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y) == ok).all()
    assert rel_l2(y[ok], y_ref[ok]) < TOL


ALL_NODES = [("das", 8, ()), ("das", 3, ()), ("mvdr", 8, ()), ("mvdr", 4, ()), ("lcmv", 8, (-60.0, 90.0)),
             ("lcmv", 16, (-60.0, 90.0, 150.0)), ("gss", 8, (-60.0,)), ("phase", 8, ()), ("phasempf", 8, ()), ("mcra", 2, ())]
# the periods jackd is usually started with besides 512 get every node; the far ends of the range the nodes of the metric, the
# recursive mask node and one 16-microphone lcmv (N = 8192 runs the in-place transforms, N = 128 / 256 blocks with idle threads)
FEW_NODES = [("das", 8, ()), ("das", 3, ()), ("mvdr", 8, ()), ("lcmv", 16, (-60.0, 90.0, 150.0)), ("phase", 8, ()), ("phasempf", 8, ()),
             ("gss", 8, (-60.0,))]
CASES = [(hop,) + n for hop in (256, 1024) for n in ALL_NODES] + [(hop,) + n for hop in (64, 128, 2048, 4096) for n in FEW_NODES]


@pytest.mark.parametrize("hop,algo,M,interf", CASES)
def test_nodes_at_other_jack_periods(hop, algo, M, interf, das_impls):
    import oracle
    p = make_params(algo, n_mics=M, interf=interf, theta=20.0, hop=hop)
    F = {64: 96, 128: 64, 256: 48, 1024: 30, 2048: 24, 4096: 20}[hop]   # enough frames behind the covariance window / MCRA start-up
    x = make_scene(M, F, hop=hop, seed=300 + M + hop // 256)
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    _, y, Y = run(p, x)
    if algo == "das" and das_impls == "f32":  # the fused fp32 kernels (das_fused_gen.hip at these periods) dump the Hermitian part of y_fft: the part that reaches Re(ifft)
        Y_ref = 0.5 * (Y_ref + np.conj(np.roll(Y_ref[:, ::-1], 1, axis=1)))
    check(y[0], Y[0], y_ref, Y_ref)   # (das in double, the default: the fp64 pipeline with the full y_fft)


@pytest.mark.parametrize("hop", [64, 128, 256, 1024, 2048, 4096])
@pytest.mark.parametrize("algo", ["das", "mvdr", "phasempf"])
def test_streaming_callbacks_at_other_jack_periods(hop, algo):
    """bf_process_hop with nframes = the configured period, one callback at a time == batch == oracle; theta in between."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 4, 14
    p = make_params(algo, n_mics=M, theta=10.0, hop=hop)
    x = make_scene(M, F, hop=hop, seed=9)
    node = oracle.OracleNode(p)
    bf = Beamformer(p)
    ys, refs = [], []
    for t in range(F):
        if t == 6:
            node.set_theta(-50.0)
            bf.set_theta(-50.0)
        seg = np.ascontiguousarray(x[:, t * hop:(t + 1) * hop])
        refs.append(node.process(seg)[0])
        ys.append(bf.process_hop(seg))
    y, y_ref = np.concatenate(ys), np.concatenate(refs)
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y) == ok).all() and rel_l2(y[ok], y_ref[ok]) < TOL
    with pytest.raises(Exception):
        bf.process_hop(np.zeros((M, 512), np.float32)) if hop != 512 else None


def test_unsupported_period_is_refused():
    from beamform_amd.capi import BfError
    from conftest import Beamformer
    _torch()
    with pytest.raises(BfError):
        Beamformer(make_params("das", n_mics=4, hop=384))      # not a power of two
    for hop in (32, 8192):                                       # outside 64 ... 4096
        with pytest.raises(BfError):
            Beamformer(make_params("mvdr", n_mics=4, hop=hop))


@pytest.mark.parametrize("algo,M,interf", [
    ("lcmv", 8, (-60.0, 90.0, 150.0, -120.0, 45.0)),                               # K = 5 at M = 8
    ("lcmv", 16, (-60.0, 90.0, 150.0, -120.0, 45.0, -20.0, 120.0, 70.0, -90.0)),   # K = 9 at M = 16
    ("gss", 8, (-60.0, 90.0, 150.0, -120.0, 45.0)),
    ("gss", 16, (-60.0, 90.0, 150.0, -120.0, 45.0, -20.0, 120.0, 70.0, -90.0)),    # K = 9 at M = 16
])
def test_more_than_three_interferers(algo, M, interf):
    import oracle
    p = make_params(algo, n_mics=M, interf=interf, theta=20.0)
    F = 40
    x = make_scene(M, F, seed=500 + len(interf))
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    bf, y, Y = run(p, x)
    assert bf.weights().shape == (1024, M, len(interf) + 1)
    check(y[0], Y[0], y_ref, Y_ref)


def test_as_many_constraints_as_microphones_runs_without_a_parity_claim():
    """K + 1 = M: W = R^-1 C (C^H R^-1 C)^-1 degenerates to C^-H, and the steering matrix of eight directions on a 20 cm
    array is numerically singular at the low bins -- the reference's inverse() and any other solver return rounding noise
    of order 1..10 there (observed on both sides).  The node must run and stay finite where the reference does; no
    parity is claimed (include/bfcore.h, bf_set_interference)."""
    import oracle
    M, interf = 8, (-60.0, 90.0, 150.0, -120.0, 45.0, -20.0, 120.0)
    p = make_params("lcmv", n_mics=M, interf=interf, theta=20.0)
    x = make_scene(M, 40, seed=507)
    y_ref, _ = oracle.OracleNode(p).process(x)
    _, y, _ = run(p, x)
    assert np.isfinite(y[0][12 * 512:]).mean() > 0.9 and np.isfinite(y_ref[12 * 512:]).mean() > 0.9


def test_interferers_appended_at_run_time_beyond_three():
    """/theta_interference keeps appending (lcmv.cpp:282-305): 2 -> 6 interferers one callback apart, then one removed."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 16, 36
    p = make_params("lcmv", n_mics=M, interf=(-60.0, 90.0), theta=20.0)
    x = make_scene(M, F, seed=77)
    node, bf = oracle.OracleNode(p), Beamformer(p)
    script = {6: (9, 150.0), 10: (9, -120.0), 14: (9, 45.0), 18: (9, -20.0), 24: (3, 89.7)}
    ys, refs = [], []
    for t in range(F):
        if t in script:
            assert bf.set_interference(*script[t]) == node.set_interference(*script[t])
            assert np.abs(bf.weights() - node.weights()).max() < 1e-14
        seg = np.ascontiguousarray(x[:, t * 512:(t + 1) * 512])
        refs.append(node.process(seg)[0])
        ys.append(bf.process(seg))
    assert bf.weights().shape[2] == 6
    y, y_ref = np.concatenate(ys), np.concatenate(refs)
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y) == ok).all() and rel_l2(y[ok], y_ref[ok]) < TOL


@pytest.mark.parametrize("algo,M,interf,radius,band", [
    ("mvdr", 24, (), 0.2, None), ("lcmv", 20, (-60.0, 90.0), 0.2, None), ("gss", 32, (-60.0,), 0.2, None),
    # the yaml's angle_interf1..15: sixteen constraints are only separable (C^H R^-1 C invertible beyond rounding noise) on an
    # aperture of many wavelengths -- a 1.5 m array in the 3-8 kHz band; on the 20 cm arrays above both sides return 1e5-size noise
    ("lcmv", 32, tuple(-170.0 + 22.0 * k for k in range(15)), 1.5, (3000.0, 8000.0))])
def test_more_than_sixteen_microphones(algo, M, interf, radius, band):
    import oracle
    rng = np.random.default_rng(4)
    ang = np.sort(rng.uniform(-np.pi, np.pi, M))
    mics = [(float(radius * np.cos(a) * (0.6 + 0.4 * rng.random())), float(radius * np.sin(a) * (0.6 + 0.4 * rng.random()))) for a in ang]
    # P = 10 frames give a rank-10 covariance: with M > 10 only the 1.001 diagonal loading keeps R invertible (cond ~ 1e3 * M),
    # exactly as in the reference; a longer window keeps the comparison meaningful
    over = dict(freq_min=band[0], freq_max=band[1]) if band else {}
    p = make_params(algo, n_mics=M, interf=interf, theta=20.0, mics=mics, past_windows=40, **over)
    F = 70
    x = make_scene(M, F, seed=600 + M, mics=mics)
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    _, y, Y = run(p, x)
    check(y[0], Y[0], y_ref, Y_ref)


@pytest.mark.parametrize("hop", [64, 256, 1024, 4096])
def test_fused_das_at_other_periods_layouts_streams_directions_cuts(hop, das_impls):
    """The fused fp32 das kernels at the JACK periods other than 512 (das_fused.hip's group mode below 512, das_fused_wave2048_kernel at 1024,
    das_fused_gen.hip above): interleaved input, several streams, look directions, a long batch cut into runs (every run but the first
    recomputes its previous frame or group), uneven batch cuts -- against the oracle."""
    import oracle
    from beamform_amd.capi import BF_INTERLEAVED
    from conftest import Beamformer
    _torch()
    if das_impls != "f32":
        pytest.skip("about the fp32 opt-in's kernels; das in double at these periods: test_nodes_at_other_jack_periods, test_streaming_callbacks_...")
    M, S, F = 5, 2, 40
    p = make_params("das", n_mics=M, theta=-30.0, hop=hop)
    xs = np.stack([make_scene(M, F, hop=hop, seed=70 + s) for s in range(S)])
    refs = [oracle.OracleNode(p).process(xs[s])[0] for s in range(S)]
    y = Beamformer(p, n_streams=S).process(xs)
    yi = Beamformer(p, n_streams=S, layout=BF_INTERLEAVED).process(np.ascontiguousarray(xs.transpose(0, 2, 1)))
    for s in range(S):
        assert rel_l2(y[s], refs[s]) < TOL and rel_l2(yi[s], refs[s]) < TOL
        if hop >= 512:
            assert np.array_equal(y[s], yi[s])
        else:  # below 512 the two layouts run different instantiations of the group-mode kernel (unrolled / run-time pair loop)
            assert np.abs(y[s].astype(np.float64) - yi[s]).max() <= 2e-6 * np.abs(refs[s]).max()
    # look directions: every beam equals its own node
    thetas = [-60.0, 10.0, 75.0]
    bf = Beamformer(p, n_dirs=3)
    bf.set_thetas(thetas)
    yd = bf.process(xs[0])
    for d, th in enumerate(thetas):
        ref, _ = oracle.OracleNode(make_params("das", n_mics=M, theta=th, hop=hop)).process(xs[0])
        assert rel_l2(yd[d], ref) < TOL
    # a batch long enough to be cut into many runs, and the same stream in uneven batches
    Fl = {64: 6000, 256: 3000, 1024: 1200, 4096: 300}[hop]
    xl = make_scene(M, Fl, hop=hop, seed=99)
    ref, _ = oracle.OracleNode(p).process(xl)
    whole = Beamformer(p).process(xl)
    assert rel_l2(whole, ref) < TOL
    b2 = Beamformer(p)
    cuts = [0, 1, 7, Fl // 3, Fl]
    parts = np.concatenate([b2.process(np.ascontiguousarray(xl[:, a * hop:c * hop])) for a, c in zip(cuts[:-1], cuts[1:])])
    if hop >= 512:
        assert np.array_equal(parts, whole)
    else:
        # below 512 frames 1024 / N consecutive frames share one transform (das_fused.hip, group mode): which ones depends on where a batch starts, so
        # the cuts agree to the rounding of the fp32 transform (1e-7 of the signal's scale), not bit for bit
        assert np.abs(parts.astype(np.float64) - whole).max() <= 1e-6 * np.abs(whole).max()
        assert rel_l2(parts, ref) < TOL


CHILD_1024 = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import oracle
from conftest import Beamformer_f32 as Beamformer   # the register-resident das kernels of this period are the fp32 opt-in
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
res = {}
for M, F in ((8, 300), (7, 41), (12, 33), (8, 4501), (1, 9)):   # gains in LDS (<= 8 microphones, one direction) and from L2 (12); 4501 frames: many runs per stream
    p = make_params("das", n_mics=M, theta=35.0, hop=1024)
    x = make_scene(M, F, hop=1024, seed=1024 + M)
    ref, _ = oracle.OracleNode(p).process(x)
    y = Beamformer(p).process(x)
    res[f"{M}/{F}"] = rel_l2(y, ref)
    if F > 1000:  # a cut stream (carried hop and tail across batches) and interleaved input
        from beamform_amd.capi import BF_INTERLEAVED
        cut = (F // 3) * 1024
        bfc = Beamformer(p)
        yc = np.concatenate([bfc.process(np.ascontiguousarray(x[:, :cut])), bfc.process(np.ascontiguousarray(x[:, cut:]))])
        res[f"{M}/{F}/cut"] = rel_l2(yc, ref)
        res[f"{M}/{F}/interleaved"] = rel_l2(Beamformer(p, layout=BF_INTERLEAVED).process(np.ascontiguousarray(x.T)), ref)
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("split", ["3", "0"])
def test_period_1024_split_kernel_and_its_switch(split, das_impls):
    """The 1024-frame period without a spectrum dump: das_fused_wave2048_kernel (one 2048-point transform per frame on a full wavefront, tails
    through an LDS ring, run boundaries by atomics: the default = BF_DAS_SPLIT2048=3) and das_fused_gen_kernel<2048> (=0) against the oracle:
    odd and single microphone counts, many runs per stream, a cut stream, interleaved input; the switch is read once per process."""
    import json, os, subprocess, sys
    if das_impls != "f32":
        pytest.skip("fp32-only kernels (das in double at this period runs the bin pipeline: test_every_jack_period)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", CHILD_1024 % dict(root=root)], env=dict(os.environ, BF_DAS_SPLIT2048=split),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert len(res) == 7 and max(res.values()) < TOL, res


CHILD_SMALL = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import oracle
from beamform_amd.capi import BF_INTERLEAVED
from conftest import Beamformer_f32 as Beamformer   # the register-resident das kernels of this period are the fp32 opt-in
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
res = {}
# odd frame counts: partial last groups; 12 microphones: gains from L2; the long ones: many runs per stream (run boundaries completed by atomics in
# the group mode of das_fused_kernel, warm-up groups in the per-run kernels); two calls in a row: the carried hop and tail across batches
for hop, M, F in ((256, 8, 301), (128, 7, 203), (64, 12, 95), (256, 1, 9), (64, 8, 1), (256, 8, 9001), (128, 4, 9003), (64, 3, 20001)):
    p = make_params("das", n_mics=M, theta=-65.0, hop=hop)
    x = make_scene(M, F, hop=hop, seed=hop + M)
    ref, _ = oracle.OracleNode(p).process(x)
    bf = Beamformer(p)
    res[f"{hop}/{M}/{F}"] = rel_l2(bf.process(x), ref)
    if F > 1000:
        cut = (F // 3) * hop
        bfc = Beamformer(p)
        yc = np.concatenate([bfc.process(np.ascontiguousarray(x[:, :cut])), bfc.process(np.ascontiguousarray(x[:, cut:]))])
        res[f"{hop}/{M}/{F}/cut"] = rel_l2(yc, ref)
        bfi = Beamformer(p, layout=BF_INTERLEAVED)
        res[f"{hop}/{M}/{F}/interleaved"] = rel_l2(bfi.process(np.ascontiguousarray(x.T)), ref)
        continue
    # the same stream one callback at a time: every group is a lone frame plus padding lanes
    bf2, node = Beamformer(p), oracle.OracleNode(p)
    ys = [bf2.process_hop(np.ascontiguousarray(x[:, t * hop:(t + 1) * hop])) for t in range(min(F, 12))]
    res[f"{hop}/{M}/{F}/hops"] = rel_l2(np.concatenate([np.asarray(v).reshape(-1) for v in ys]), ref[:min(F, 12) * hop])
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("il", ["1", "0"])
def test_small_periods_interleaving_kernel_and_its_switch(il, das_impls):
    """Periods 256 / 128 / 64 without a spectrum dump, 1024 / N frames interleaved into one 1024-point pass: das_fused_kernel in group mode
    (BF_DAS_INTERLEAVE=1, the default) and das_fused_gen_kernel<N> (=0) against the oracle: odd frame counts, one callback at a time,
    12 microphones, many runs per stream, a cut stream, interleaved input."""
    import json, os, subprocess, sys
    if das_impls != "f32":
        pytest.skip("fp32-only kernels (das in double at these periods runs the bin pipeline: test_every_jack_period)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", CHILD_SMALL % dict(root=root)], env=dict(os.environ, BF_DAS_INTERLEAVE=il),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert len(res) == 19 and max(res.values()) < TOL, res


CHILD_STFT = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import oracle
from beamform_amd.capi import Beamformer, BF_INTERLEAVED
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
res = {}
for algo, hop, M, F in (("mvdr", 256, 8, 45), ("phasempf", 128, 5, 33), ("gss", 64, 3, 41), ("mvdr", 1024, 7, 25), ("phase", 1024, 2, 9), ("mcra", 256, 1, 30),
                        ("phase", 256, 8, 77), ("phase", 64, 3, 51), ("phasempf", 256, 8, 40), ("phase", 128, 7, 1),
                        ("phase", 1024, 8, 19), ("phasempf", 1024, 3, 12), ("phase", 1024, 5, 1)):
    interf = (-60.0,) if algo == "gss" else ()
    p = make_params(algo, n_mics=M, theta=35.0, hop=hop, interf=interf)
    x = make_scene(M, F, hop=hop, seed=hop + M)
    ref, _ = oracle.OracleNode(p).process(x)
    ok = np.isfinite(ref)
    y = Beamformer(p).process(x)
    yi = Beamformer(p, layout=BF_INTERLEAVED).process(np.ascontiguousarray(x.T))
    res[f"{algo}/{hop}/{M}"] = max(rel_l2(y[ok], ref[ok]), rel_l2(yi[ok], ref[ok])) if (np.isfinite(y) == ok).all() and (np.isfinite(yi) == ok).all() else 1.0
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("env", [{}, {"BF_FUSED_BINS": "0"}, {"BF_STFT_SMALL": "0", "BF_STFT_SPLIT": "0", "BF_FUSED_BINS": "0"}],
                         ids=["registers", "registers-unfused", "generic"])
def test_fp64_nodes_stft_kernels_at_other_periods_and_their_switches(env, das_impls):
    """stft_small_kernel / istft_small_kernel (N = 128 / 256 / 512: several frames per half-wavefront through one transpose plane),
    stft_wave2048_kernel (N = 2048: one transform per full wavefront) /
    istft_split_kernel, stft_bins_small_kernel / stft_bins_split_kernel (phase / phasempf below 512 / at 1024: the STFT and
    the per-bin stage in one launch; BF_FUSED_BINS=0: the chain) and the generic LDS-staged kernels they replace (BF_STFT_SMALL=0 / BF_STFT_SPLIT=0),
    against the oracle: odd microphone counts, one microphone, both layouts, frame counts that leave partial groups, rounds and short runs."""
    import json, os, subprocess, sys
    if das_impls != "f64":
        pytest.skip("no das in this test: once is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", CHILD_STFT % dict(root=root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert len(res) == 13 and max(res.values()) < TOL, res
