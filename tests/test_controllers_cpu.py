"""Theta controllers: beamform_amd/controllers.py against the committed fixture
tests/golden/controllers_loop.npz (window stream in, published theta sequences out; tests/golden/make_controllers_golden.py) and
against oracle/controllers_oracle.py, the restatement of the reference scripts' callbacks that produced it.  The published
angles must be identical, value for value.  The closed loop (`controllers.follow`) is then run around the oracle node."""
import json
import math
import os

import numpy as np
import pytest

import oracle
from oracle import controllers_oracle as co
from beamform_amd import controllers
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "controllers_loop.npz")


@pytest.fixture(scope="module")
def golden():
    g = np.load(GOLDEN)
    sc = json.loads(str(g["scene"]))
    as_msg = lambda a: tuple(float(v) for v in a)   # rospy hands the callbacks float32[] fields as tuples of Python floats
    ys, refs = [as_msg(w) for w in g["win"]], [as_msg(r) for r in g["ref"]]
    seq = {n: list(zip(g["k_" + n].tolist(), g["theta_" + n].tolist())) for n in ("energy", "diff", "spec_history", "spec_spectrogram")}
    return sc, ys, refs, seq


def test_fixture_windows_are_the_oracle_nodes_output(golden):
    """The stored window stream is what the CPU oracle's das node emits for the stored scene parameters (drift check)."""
    sc, ys, refs, _ = golden
    p = make_params("das", n_mics=sc["n_mics"], theta=sc["node_theta"])
    x = make_scene(sc["n_mics"], sc["n_frames"], seed=sc["seed"], theta_s=sc["theta_s"], silent_frac=sc["silent_frac"])
    y, _ = oracle.OracleNode(p).process(x)
    assert np.array_equal(y.reshape(-1, 512).astype(np.float32), np.array(ys, np.float32))
    assert np.array_equal(x[0].reshape(-1, 512), np.array(refs, np.float32))


def test_oracle_restatement_reproduces_the_fixture(golden):
    sc, ys, refs, seq = golden
    pairs = list(zip(ys, refs))
    assert co.ref_energy2theta(ys, sc["node_theta"]) == seq["energy"]
    assert co.ref_energy2theta_diff(pairs, sc["node_theta"]) == seq["diff"]
    assert co.ref_energy2theta_spec(pairs, sc["node_theta"], method="history") == seq["spec_history"]
    assert co.ref_energy2theta_spec(pairs, sc["node_theta"], method="spectrogram", num_win=30) == seq["spec_spectrogram"]


def test_energy2theta_matches_the_fixture(golden):
    sc, ys, _, seq = golden
    ctl = controllers.Energy2Theta(initial_angle=sc["node_theta"])
    got = [(k, th) for k, w in enumerate(ys) if (th := ctl.on_window(w)) is not None]
    want = seq["energy"]
    assert len(want) > 60 and got == want
    assert want[0][0] == 50                       # fifty active windows fill the deque before the first step
    assert all(k <= 151 for k, _ in want)         # the silent tail (from hop 150, one hop of latency) is gated out by the VAD threshold


def test_energy2theta_diff_and_spec_match_the_fixture(golden):
    sc, ys, refs, seq = golden
    pairs = list(zip(ys, refs))
    ctl = controllers.Energy2ThetaDiff(initial_angle=sc["node_theta"])
    got = [(k, th) for k, (a, r) in enumerate(pairs) if (th := ctl.on_windows(a, r)) is not None]
    assert len(seq["diff"]) > 100 and got == seq["diff"]
    ctl = controllers.Energy2ThetaSpec(initial_angle=sc["node_theta"], num_win=100, method="history")
    got = [(k, th) for k, (a, r) in enumerate(pairs) if (th := ctl.on_windows(a, r)) is not None]
    assert len(seq["spec_history"]) > 30 and got == seq["spec_history"]
    # energy_calc_method = 'spectrogram' (energy2theta-spec.py:55-75): sqrt of the mean thresholded spectrogram bin of the deque, mu = 5000
    ctl = controllers.Energy2ThetaSpec(initial_angle=sc["node_theta"], num_win=30, method="spectrogram")
    got = [(k, th) for k, (a, r) in enumerate(pairs) if (th := ctl.on_windows(a, r)) is not None]
    assert len(seq["spec_spectrogram"]) > 100 and got == seq["spec_spectrogram"]
    assert np.ptp([t for _, t in got]) > 0.5      # mu = 5000 really moves the angle


def test_spectrogram_energy_below_the_threshold_is_invalid():
    """An empty thresholded spectrogram gives np.mean([]) = nan -> -100 ('invalid', energy2theta-spec.py:100-101): no step."""
    ctl = controllers.Energy2ThetaSpec(initial_angle=5.0, num_win=4, method="spectrogram", vad_threshold=0.0)
    quiet = np.full(512, 1e-6)
    assert [ctl.on_windows(np.zeros(512), quiet) for _ in range(8)] == [None] * 8
    assert co.ref_energy2theta_spec([(tuple(np.zeros(512)), tuple(quiet))] * 8, 5.0, method="spectrogram", num_win=4, vad_threshold=0.0) == []


def test_wrap_and_sir2theta():
    assert controllers.wrap180(190.0) == -170.0 and controllers.wrap180(-181.0) == 179.0 and controllers.wrap180(30.0) == 30.0
    c = controllers.SIR2Theta()
    want = [t for _, t in co.ref_sir2theta((3.0, 4.5, 2.0))]   # SIR2theta.py:9-26
    assert c.initial() == 1.0 and [c.on_sir(s) for s in (3.0, 4.5, 2.0)] == want
    big = controllers.Energy2Theta(initial_angle=170.0, num_win=1, mu=1e6)
    w = np.full(512, 0.1)
    assert big.on_window(w) is None and big.on_window(w) == 170.0     # energy unchanged: no movement
    assert -180.0 <= controllers.wrap180(170.0 + 30.0) <= 180.0


def test_vad_states():
    v = controllers.Vad()
    quiet, loud = np.full(512, 0.001), np.full(512, 0.2)
    seq = [v.on_window(quiet) for _ in range(10)] + [v.on_window(loud) for _ in range(3)] + [v.on_window(quiet) for _ in range(10)]
    assert not any(seq[:10]) and any(seq[10:13]) and not any(seq[-5:])


def test_closed_loop_around_the_oracle_node():
    """controllers.follow: process_hop -> controller -> set_theta, one period at a time; the angles it publishes are the
    transcription's angles for the windows that loop produced (the loop feeds back, so the windows are its own)."""
    M, F = 4, 140
    p = make_params("das", n_mics=M, theta=-15.0)
    x = make_scene(M, F, seed=43, theta_s=20.0, silent_frac=0.0)
    node = oracle.OracleNode(p)
    y, pub = controllers.follow(node, x, controllers.Energy2Theta(initial_angle=-15.0, num_win=30, mu=25.0))
    assert len(pub) == F - 30
    want = co.ref_energy2theta([tuple(float(v) for v in y[t * 512:(t + 1) * 512]) for t in range(F)], -15.0, num_win=30)
    assert pub == want
    assert np.ptp([t for _, t in pub]) > 0        # the angle really moves


def test_jack_ref_is_the_input_delayed_by_one_hop():
    """jack_ref.cpp:19-30: mic 0 through the sqrt-Hann^2 WOLA with no processing = the input one hop later (float rounding)."""
    rng = np.random.default_rng(3)
    s = (rng.standard_normal(20 * 512) * 0.2).astype(np.float32)
    ref = controllers.JackRef(512)
    out = np.concatenate([ref.process_hop(s[t * 512:(t + 1) * 512]) for t in range(20)])
    assert np.array_equal(out[:512], np.zeros(512, np.float32))
    assert np.abs(out[512:] - s[:-512]).max() < 2e-7


def test_diff_controller_sees_an_aligned_reference():
    """energy2theta-diff.py:74 subtracts the beamformer output from jackaudio_ref window by window.  With identical
    channels on co-located microphones (every steering delay 0) das returns the input through the same WOLA, so the aligned
    difference is ~0 -- with the undelayed reference channel it would be the full-scale difference of two hops."""
    M, F = 4, 40
    p = make_params("das", n_mics=M, mics=[(0.0, 0.0)] * M)
    s = (np.random.default_rng(5).standard_normal(F * 512) * 0.2).astype(np.float32)
    x = np.tile(s, (M, 1))
    seen = []

    class Spy(controllers.Energy2ThetaDiff):
        def on_windows(self, win, win_ref):
            seen.append(float(np.abs(np.asarray(win_ref, np.float64) - np.asarray(win, np.float64)).max()))
            return super().on_windows(win, win_ref)

    controllers.follow(oracle.OracleNode(p), x, Spy(initial_angle=0.0))
    assert len(seen) == F and max(seen) < 1e-6, max(seen)
