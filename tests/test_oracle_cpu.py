"""CPU suite (no GPU): the oracle against its golden vectors, an independent numpy restatement and
known-answer properties that follow from the reference source.

PARITY UNPINNED: the reference ships no tests/golden vectors and cannot be built here (ROS, JACK, FFTW3,
Eigen3 absent), so the strongest pin available is two independent restatements + these known answers."""
import glob
import json
import os

import numpy as np
import pytest

import oracle
from oracle import np_oracle
from beamform_amd.params import AIRA16_XY, make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
              if not os.path.basename(p).startswith(("wav_", "controllers_", "resample_")))   # those belong to test_wavio_*, test_controllers_*, test_resample_*


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_oracle_reproduces_golden(path):
    d = np.load(path)
    p = json.loads(str(d["params"]))
    p["mics"] = [tuple(m) for m in p["mics"]]
    y, Y = oracle.OracleNode(p).process(d["x"], want_spectrum=True)
    assert np.array_equal(y, d["y"], equal_nan=True)
    fin = np.isfinite(d["Y"]).all(axis=1)
    assert (np.isfinite(Y).all(axis=1) == fin).all()
    assert max(rel_l2(Y[t], d["Y"][t]) for t in range(len(fin)) if fin[t]) < 1e-12


@pytest.mark.parametrize("algo,M,interf", [("das", 4, ()), ("mvdr", 8, ()), ("lcmv", 8, (-60.0, 90.0)),
                                           ("gss", 8, (-60.0,)), ("gss", 3, ()), ("phase", 8, ()), ("phasempf", 8, ()),
                                           ("mcra", 4, ())])
def test_oracle_agrees_with_numpy_restatement(algo, M, interf):
    p = make_params(algo, n_mics=M, interf=interf, theta=-30.0)
    x = make_scene(M, 20, seed=17)
    y, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    y2, Y2 = np_oracle.process(p, x)
    fin = np.isfinite(Y).all(axis=1)
    assert (np.isfinite(Y2).all(axis=1) == fin).all()
    assert max(rel_l2(Y[t], Y2[t]) for t in range(20) if fin[t]) < 1e-10
    ok = np.isfinite(y)
    assert np.array_equal(y[ok], y2[ok])


def test_mcra_node_minima_reset_agrees_with_numpy_restatement():
    """mcra.cpp:100-113 with a short L so the S_min/S_tmp reset and the end of first_L fall inside the run."""
    p = make_params("mcra", n_mics=2, mcra_L=6)
    x = make_scene(2, 40, seed=23)
    y, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    y2, Y2 = np_oracle.process(p, x)
    assert max(rel_l2(Y[t], Y2[t]) for t in range(40)) < 1e-10
    assert np.array_equal(y, y2)
    assert np.all(Y[:, 0] == 0)           # Q16: bin 0 never written
    assert np.abs(Y[5:, 1:]).max() > 0    # and the rest is live


@pytest.mark.parametrize("M,F,over", [(4, 5, {}), (8, 4, dict(gsc_filter_size=32)), (3, 5, dict(gsc_use_vad=1, gsc_vad_threshold=0.15))])
def test_gsc_agrees_with_numpy_restatement(M, F, over):
    """gsc.cpp:54-197: both restatements replay the float32 NLMS operation by operation."""
    p = make_params("gsc", n_mics=M, theta=20.0, **over)
    x = make_scene(M, F, seed=3)
    y = oracle.OracleNode(p).process(x)[0]
    y2 = np_oracle.process(p, x)[0]
    assert np.isfinite(y).all() and np.abs(y).max() > 0.1
    assert np.abs(y - y2).max() < 1e-6 * np.abs(y).max()
    yd = oracle.OracleNode(make_params("das", n_mics=M, theta=20.0)).process(x)[0]
    assert rel_l2(y, yd) > 1e-3  # the sidelobe canceller does something


def test_fft_matches_numpy():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    assert rel_l2(oracle.fft(x, -1), np.fft.fft(x)) < 1e-14
    assert rel_l2(oracle.fft(x, +1), np.fft.ifft(x) * 1024) < 1e-14


def test_frequency_vector_known_answers():
    """SURVEY App. C probe 1 (util.h:190-199 run behind a shim by the survey): quirk Q1."""
    f = oracle.OracleNode(make_params("das", n_mics=4)).freqs()
    assert f[0] == 0.0 and f[1] == 46.875 and f[510] == 23906.25
    assert f[511] == 24000.0          # overwritten with sr/2
    assert f[512] == 0.0              # never written by the reference; defined 0
    assert f[513] == -23953.125 and f[1023] == -46.875
    assert np.array_equal(f, np_oracle.freq_vector(1024, 48000.0))


def test_wola_identity_known_answer():
    """jack_ref.cpp / util.h:301-302: sqrt-Hann^2 at 50 % hop reconstructs the input delayed by one hop.
    Co-located mics at theta = 0 make every steering weight 1, so DAS is the identity path."""
    M, F = 3, 8
    p = make_params("das", n_mics=M, mics=[(0.0, 0.0)] * M)
    x = np.ones((M, F * 512), np.float32)
    y, _ = oracle.OracleNode(p).process(x)
    assert np.abs(y[:512]).max() < 1e-6     # hop 0 = first half of frame 0 = the pre-filled zero hop
    assert np.abs(y[512:] - 1.0).max() < 1e-6
    rng = np.random.default_rng(1)
    s = rng.uniform(-0.5, 0.5, F * 512).astype(np.float32)
    y, _ = oracle.OracleNode(p).process(np.tile(s, (M, 1)))
    assert np.abs(y[512:] - s[:-512]).max() < 1e-6


def test_delays_and_weights_known_answers():
    """util.h:143-159: tau_0 = 0; tau_m = dist*cos(angle_m - theta)/(-343) from the RAW coordinates (Q2)."""
    mics = AIRA16_XY[:4]
    node = oracle.OracleNode(make_params("das", n_mics=4, theta=30.0))
    d = node.delays()
    assert d[0] == 0.0
    for m in range(1, 4):
        x, y = mics[m]
        expect = np.hypot(x, y) * np.cos(np.radians(np.degrees(np.arctan2(y, x)) - 30.0)) / -343.0
        assert abs(d[m] - expect) < 1e-15
    w = node.weights()[:, :, 0]
    assert np.all(w[:, 0] == 1.0)                                    # row 0 = 1 (das.cpp:33-38)
    assert np.abs(np.abs(w) - 1.0).max() < 1e-15
    f = node.freqs()
    assert np.abs(w[5, 2] - np.exp(-2j * np.pi * f[5] * d[2])).max() < 1e-15
    # the angle difference is wrapped into [-180, 180] before the cosine (util.h:151-155)
    far = oracle.OracleNode(make_params("das", n_mics=4, theta=-170.0)).delays()
    assert np.all(np.isfinite(far))


def test_das_zero_delay_is_channel_mean():
    """das.cpp:60-63 with all weights 1: Y = mean_m X_m."""
    M = 4
    p = make_params("das", n_mics=M, mics=[(0.0, 0.0)] * M)
    x = make_scene(M, 6, seed=3)
    _, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    X = np_oracle.stft(p, x)
    assert rel_l2(Y, X.mean(axis=1)) < 1e-14


def test_mvdr_frame0_is_nan_and_constraint_holds():
    """mvdr.cpp:87-94: the covariance excludes the current frame, so frame 0 inverts the zero matrix (NaN/Inf,
    SURVEY A.3).  The weights the reference forms satisfy w^H a = 1: checked on the numpy restatement's maths."""
    M = 8
    p = make_params("mvdr", n_mics=M, theta=20.0)
    x = make_scene(M, 14, seed=5, silent_frac=0.0)
    y, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    assert not np.isfinite(Y[0]).all() and np.isfinite(Y[1:]).all()
    assert not np.isfinite(y[:1024]).all() and np.isfinite(y[1024:]).all()
    X = np_oracle.stft(p, x)
    a = np_oracle.steering(p, 20.0)[:, 100]
    hist = X[2:12, :, 100].T
    R = (hist @ hist.conj().T) * (np.ones((M, M)) + 0.001 * np.eye(M))
    Ri = np.linalg.inv(R)
    w = (Ri @ a) / (a.conj() @ Ri @ a)
    assert abs(w.conj() @ a - 1.0) < 1e-9


def test_lcmv_without_interferers_equals_mvdr_except_bin0():
    """lcmv.cpp:102 loops from j = 0 (bin 0 is out of band -> 0) where mvdr.cpp:76 copies in_fft(0,0)."""
    M = 4
    x = make_scene(M, 14, seed=6)
    _, Ym = oracle.OracleNode(make_params("mvdr", n_mics=M)).process(x, want_spectrum=True)
    _, Yl = oracle.OracleNode(make_params("lcmv", n_mics=M)).process(x, want_spectrum=True)
    assert np.all(Yl[:, 0] == 0)
    assert rel_l2(Yl[1:, 1:], Ym[1:, 1:]) < 1e-9


def test_out_of_band_bins_are_zero_and_gate_branch():
    M = 4
    p = make_params("mvdr", n_mics=M)
    x = make_scene(M, 20, seed=8, silent_frac=0.3)
    _, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    f = np.abs(oracle.OracleNode(p).freqs())
    oob = (f < 100.0) | (f > 16000.0)
    oob[0] = False
    assert np.all(Y[1:, oob] == 0)
    X = np_oracle.stft(p, x)
    quiet = (np.abs(X).sum(axis=1) / (M * 1024)) <= 0.001          # gate closed -> 0.01 * mic 0
    sel = quiet[16:, :] & ~oob[None, :]
    sel[:, 0] = False
    assert sel.any()
    assert np.abs(Y[16:][sel] - 0.01 * X[16:, 0, :][sel]).max() < 1e-15


def test_phase_identical_channels_pass_unattenuated():
    """phase.cpp:102-118: identical signals + zero delays -> all aligned phases equal -> mean difference 0 < threshold."""
    M = 4
    p = make_params("phase", n_mics=M, mics=[(0.0, 0.0)] * M, mag_threshold=0.0)
    s = make_scene(1, 8, seed=9)[0]
    x = np.tile(s, (M, 1))
    _, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    X = np_oracle.stft(p, x)[:, 0, :]
    assert rel_l2(Y[:, 1:], X[:, 1:]) < 1e-12


def test_streaming_hop_by_hop_equals_batch():
    """config 1: das 4-mic single-frame streaming == batch (the oracle's process() is a loop of callbacks)."""
    M, F = 4, 10
    p = make_params("das", n_mics=M)
    x = make_scene(M, F, seed=4)
    y, _ = oracle.OracleNode(p).process(x)
    node = oracle.OracleNode(p)
    y2 = np.concatenate([node.process_hop(x[:, t * 512:(t + 1) * 512])[0] for t in range(F)])
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("hop", [256, 1024])
@pytest.mark.parametrize("algo,M,interf", [("das", 4, ()), ("mvdr", 4, ()), ("lcmv", 8, (-60.0, 90.0, 150.0, -120.0, 45.0)),
                                           ("phasempf", 3, ())])
def test_other_jack_periods_and_more_interferers_agree_with_numpy_restatement(hop, algo, M, interf):
    """rosjack.cpp:131 takes whatever period the JACK server runs (fft_win = 2 * period, util.h:261); lcmv.cpp:258-309 appends
    interferers without a cap.  The two restatements must agree there too."""
    p = make_params(algo, n_mics=M, interf=interf, theta=25.0, hop=hop)
    F = 16 if hop == 256 else 14
    x = make_scene(M, F, hop=hop, seed=31)
    y, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
    assert Y.shape == (F, 2 * hop)
    y2, Y2 = np_oracle.process(p, x)
    fin = np.isfinite(Y).all(axis=1)
    assert (np.isfinite(Y2).all(axis=1) == fin).all()
    # six constraints on eight microphones: C^H R^-1 C is far worse conditioned than the launch-file shapes, and the two
    # restatements invert it differently (own LU vs LAPACK): agreement to 1e-7 instead of 1e-12, float output to the last bits
    loose = len(interf) > 3
    assert max(rel_l2(Y[t], Y2[t]) for t in range(F) if fin[t]) < (1e-6 if loose else 1e-9)
    ok = np.isfinite(y)
    if loose:
        assert rel_l2(y[ok], y2[ok]) < 1e-6
    else:
        assert np.array_equal(y[ok], y2[ok])
