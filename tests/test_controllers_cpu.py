"""Theta controllers against a numpy transcription of the reference scripts' callbacks.

The transcriptions below restate scripts/energy2theta.py:62-101 (+ :23-60), energy2theta-diff.py:72-103 (+ :60),
energy2theta-spec.py:105-150 (+ :39-103, 'history' method) and SIR2theta.py:9-26 with their module-level globals as a dict
(test infrastructure, like the oracle).  Both sides are driven by the same oracle output windows; the published angles
must be identical.  The closed loop (`controllers.follow`) is then run around the oracle node."""
import math
from collections import deque

import numpy as np
import pytest

import oracle
from beamform_amd import controllers
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene


# ---- transcriptions ---------------------------------------------------------------------------------------------------
def ref_energy_from_list(data_list):                              # energy2theta.py:23-27
    sq = [i ** 2 for i in data_list]
    return math.sqrt(sum(sq) / len(sq))


def ref_energy2theta(windows_in, initial_angle, num_win=50, vad_threshold=0.001, mu=25):
    g = dict(num_win_i=0, past_energy=-100.0, past_theta=initial_angle, windows=deque([]), hist_bins=[])
    out = []

    def energy_from_deque(dq):                                    # energy2theta.py:29-60
        data_list = [item for sub in list(dq) for item in sub]
        data_np = np.abs(np.array(data_list))
        if len(g["hist_bins"]) > 0:
            vals, bins = np.histogram(data_np, g["hist_bins"])
        else:
            vals, bins = np.histogram(data_np, "fd")
            g["hist_bins"] = bins
        p = vals.astype(float) / len(data_list)
        return np.sum(bins[0:-1] * p)

    for k, data in enumerate(windows_in):                         # energy2theta.py:62-101
        this_win = list(data)
        if ref_energy_from_list(this_win) >= vad_threshold:
            if g["num_win_i"] < num_win:
                g["windows"].append(this_win)
                g["num_win_i"] += 1
            else:
                g["windows"].popleft()
                g["windows"].append(this_win)
                if g["past_energy"] == -100.0:
                    g["past_energy"] = energy_from_deque(g["windows"])
                energy = energy_from_deque(g["windows"])
                theta = g["past_theta"] + mu * (energy - g["past_energy"])
                if theta > 180:
                    theta = theta - 360
                elif theta < -180:
                    theta = theta + 360
                out.append((k, theta))
                g["past_energy"], g["past_theta"] = energy, theta
    return out


def ref_energy2theta_diff(pairs, initial_angle, num_win=50, vad_threshold=0.001, mu=25):
    g = dict(num_win_i=0, past_energy=-100.0, past_theta=initial_angle, windows=deque([]))
    out = []
    for k, (a, r) in enumerate(pairs):                            # energy2theta-diff.py:72-103
        this_win = (np.array(list(r)) - np.array(list(a))).tolist()
        if g["num_win_i"] < num_win:
            g["windows"].append(this_win)
            g["num_win_i"] += 1
        else:
            g["windows"].popleft()
            g["windows"].append(this_win)
        if ref_energy_from_list(this_win) >= vad_threshold:
            def e():                                              # :31-62: RMS of the deque
                d = np.abs(np.array([item for sub in list(g["windows"]) for item in sub]))
                return math.sqrt(np.mean(d ** 2))
            if g["past_energy"] == -100.0:
                g["past_energy"] = e()
            energy = e()
            theta = g["past_theta"] - mu * (energy - g["past_energy"])
            if theta > 180:
                theta = theta - 360
            elif theta < -180:
                theta = theta + 360
            out.append((k, theta))
            g["past_energy"], g["past_theta"] = energy, theta
    return out


def ref_energy2theta_spec_history(pairs, initial_angle, num_win=100, vad_threshold=0.001):
    g = dict(num_win_i=0, past_energy=-100.0, past_theta=initial_angle, windows=deque([]), mu=5000)
    out = []

    def e():                                                      # energy2theta-spec.py:39-103, 'history'
        g["mu"] = 10
        alpha = 1000
        past_values = np.array([np.sqrt(np.mean(np.array(w) ** 2)) for w in list(g["windows"])])
        delta = past_values[-1] - np.mean(past_values)
        with np.errstate(divide="ignore", invalid="ignore"):
            energy = past_values[-1] / (delta * alpha)
        return -100.0 if math.isnan(energy) else energy

    for k, (a, r) in enumerate(pairs):                            # :105-150
        this_win = (np.array(list(r)) - np.array(list(a))).tolist()
        if g["num_win_i"] < num_win:
            g["windows"].append(this_win)
            g["num_win_i"] += 1
        else:
            if g["num_win_i"] == num_win:
                g["num_win_i"] += 1
            g["windows"].popleft()
            g["windows"].append(this_win)
            if ref_energy_from_list(this_win) >= vad_threshold:
                if g["past_energy"] == -100.0:
                    g["past_energy"] = e()
                energy = e()
                if energy > -100.0:
                    theta = g["past_theta"] + g["mu"] * (energy - g["past_energy"])
                    if theta > 180:
                        theta = theta - 360
                    elif theta < -180:
                        theta = theta + 360
                    out.append((k, theta))
                    g["past_energy"], g["past_theta"] = energy, theta
    return out


# ---- data -------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def windows():
    """Output windows of an oracle das node looking 35 degrees off the source, plus the reference channel."""
    M, F = 4, 170
    p = make_params("das", n_mics=M, theta=-15.0)
    x = make_scene(M, F, seed=41, theta_s=20.0, silent_frac=0.12)
    y, _ = oracle.OracleNode(p).process(x)
    # rospy hands the callbacks float32[] fields as tuples of Python floats: the transcriptions get exactly that
    as_msg = lambda a: tuple(float(v) for v in a)
    return [as_msg(y[t * 512:(t + 1) * 512]) for t in range(F)], [as_msg(x[0, t * 512:(t + 1) * 512]) for t in range(F)]


def test_energy2theta_matches_the_script(windows):
    ys, _ = windows
    ctl = controllers.Energy2Theta(initial_angle=-15.0)
    got = [(k, th) for k, w in enumerate(ys) if (th := ctl.on_window(w)) is not None]
    want = ref_energy2theta(ys, -15.0)
    assert len(want) > 60 and len(got) == len(want)
    assert [k for k, _ in got] == [k for k, _ in want]
    assert np.array_equal([t for _, t in got], [t for _, t in want])
    assert want[0][0] == 50                       # fifty active windows fill the deque before the first step
    assert all(k <= 151 for k, _ in want)         # the silent tail (from hop 150, one hop of latency) is gated out by the VAD threshold


def test_energy2theta_diff_and_spec_match_the_scripts(windows):
    ys, refs = windows
    pairs = list(zip(ys, refs))
    ctl = controllers.Energy2ThetaDiff(initial_angle=-15.0)
    got = [(k, th) for k, (a, r) in enumerate(pairs) if (th := ctl.on_windows(a, r)) is not None]
    want = ref_energy2theta_diff(pairs, -15.0)
    assert len(want) > 100 and got == want
    ctl = controllers.Energy2ThetaSpec(initial_angle=-15.0, num_win=100, method="history")
    got = [(k, th) for k, (a, r) in enumerate(pairs) if (th := ctl.on_windows(a, r)) is not None]
    want = ref_energy2theta_spec_history(pairs, -15.0)
    assert len(want) > 30 and got == want
    spec = controllers.Energy2ThetaSpec(initial_angle=0.0, num_win=20, method="spectrogram")
    th = [spec.on_windows(a, r) for a, r in pairs[:60]]
    assert all(t is None for t in th[:20]) and any(t is not None and math.isfinite(t) for t in th[20:])


def test_wrap_and_sir2theta():
    assert controllers.wrap180(190.0) == -170.0 and controllers.wrap180(-181.0) == 179.0 and controllers.wrap180(30.0) == 30.0
    c = controllers.SIR2Theta()
    past_sir, past_theta, want = -100.0, 1.0, []
    for sir in (3.0, 4.5, 2.0):                   # SIR2theta.py:9-26
        theta = past_theta - 0.01 * (sir - past_sir)
        want.append(theta)
        past_sir, past_theta = sir, theta
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
    want = ref_energy2theta([tuple(float(v) for v in y[t * 512:(t + 1) * 512]) for t in range(F)], -15.0, num_win=30)
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
