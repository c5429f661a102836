"""Steering controllers: the host-side mirror of the reference's /theta publishers.

The reference closes the loop outside the node: a Python script subscribes to the node's output windows
(`jackaudio`, one message per JACK period), estimates an energy over the last `num_win` windows and publishes a new
/theta by a gradient step on that energy (scripts/energy2theta.py:62-101 and its -diff / -spec variants,
scripts/SIR2theta.py:9-26).  These classes keep that behaviour -- same gate, same window deque, same energy
estimators, same step and wrap rule -- as plain objects with one method per ROS callback, so the loop can run around a
`Beamformer` (`follow`) one period at a time, exactly as the topics would drive it.  Control plane only: the data path
stays in the HIP kernels; `beamform_amd.capi.Beamformer.stream_rms` (bf_stream_rms) is the same
`get_energy_from_list` quantity evaluated on the GPU when the windows stay resident (examples/theta_scan.cpp).

Python because the reference's controllers are Python; numpy/scipy only.
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np

INVALID = -100.0   # the scripts' "no energy yet" sentinel (energy2theta.py:17)


def window_rms(win) -> float:
    """get_energy_from_list (energy2theta.py:23-27): root mean square of one output window."""
    w = np.asarray(win, dtype=np.float64)
    return math.sqrt(float(np.sum(w * w)) / len(w))


def wrap180(theta: float) -> float:
    """energy2theta.py:85-88: one wrap step, not a modulo (a step larger than 360 degrees stays out of range, as there)."""
    if theta > 180:
        return theta - 360
    if theta < -180:
        return theta + 360
    return theta


class Energy2Theta:
    """scripts/energy2theta.py: gradient ASCENT on the expected |sample| of the last `num_win` active windows.

    on_window(win) is energycallback (:62-101): windows below the VAD threshold are ignored altogether; the first
    `num_win` active windows only fill the deque; from then on every active window slides the deque, estimates the
    energy and publishes theta = past_theta + mu * (energy - past_energy), wrapped once into [-180, 180].
    The energy (get_energy_from_deque, :29-60) is the expectation of |x| under a histogram whose bin edges are chosen by
    numpy's Freedman-Diaconis rule on the FIRST evaluation and frozen afterwards (left bin edges x relative counts)."""

    def __init__(self, initial_angle: float = 0.0, num_win: int = 50, vad_threshold: float = 0.001, mu: float = 25.0):
        self.num_win, self.vad_threshold, self.mu = int(num_win), float(vad_threshold), float(mu)
        self.past_theta = float(initial_angle)      # rospy.get_param('/beamform/initial_angle') (:109-112)
        self.past_energy = INVALID
        self.windows = deque()
        self.hist_bins = None
        self.n_published = 0

    def deque_energy(self) -> float:
        data = np.abs(np.concatenate([np.asarray(w, dtype=np.float64) for w in self.windows]))
        if self.hist_bins is None:
            counts, edges = np.histogram(data, "fd")
            self.hist_bins = edges
        else:
            counts, edges = np.histogram(data, self.hist_bins)
        p = counts.astype(float) / len(data)
        return float(np.sum(edges[0:-1] * p))

    def step(self, energy: float) -> float:
        theta = wrap180(self.past_theta + self.mu * (energy - self.past_energy))
        self.past_energy, self.past_theta = energy, theta
        self.n_published += 1
        return theta

    def on_window(self, win):
        """One `jackaudio` message.  Returns the theta to publish, or None."""
        if window_rms(win) < self.vad_threshold:
            return None
        if len(self.windows) < self.num_win:
            self.windows.append(np.array(win, dtype=np.float64))
            return None
        self.windows.popleft()
        self.windows.append(np.array(win, dtype=np.float64))
        if self.past_energy == INVALID:
            self.past_energy = self.deque_energy()
        return self.step(self.deque_energy())


class Energy2ThetaDiff:
    """scripts/energy2theta-diff.py: gradient DESCENT on the RMS of (reference channel - beamformer output) over the last
    `num_win` windows (:72-103).  Unlike energy2theta the deque slides on every message; only the step is VAD-gated, and the
    energy is the plain RMS of the deque (:60)."""

    def __init__(self, initial_angle: float = 0.0, num_win: int = 50, vad_threshold: float = 0.001, mu: float = 25.0):
        self.num_win, self.vad_threshold, self.mu = int(num_win), float(vad_threshold), float(mu)
        self.past_theta = float(initial_angle)
        self.past_energy = INVALID
        self.windows = deque()

    def deque_energy(self) -> float:
        data = np.abs(np.concatenate(list(self.windows)))
        return math.sqrt(float(np.mean(data ** 2)))

    def on_windows(self, win, win_ref):
        """One synchronised (`jackaudio`, `jackaudio_ref`) pair.  Returns the theta to publish, or None."""
        d = np.asarray(win_ref, dtype=np.float64) - np.asarray(win, dtype=np.float64)
        if len(self.windows) >= self.num_win:
            self.windows.popleft()
        self.windows.append(d)
        if window_rms(d) < self.vad_threshold:
            return None
        if self.past_energy == INVALID:
            self.past_energy = self.deque_energy()
        energy = self.deque_energy()
        theta = wrap180(self.past_theta - self.mu * (energy - self.past_energy))
        self.past_energy, self.past_theta = energy, theta
        return theta


class Energy2ThetaSpec:
    """scripts/energy2theta-spec.py (:105-150): as -diff, but the deque must be full before anything else happens, the
    step is an ascent again, and the energy is one of
      'history'     (:77-94)  last window's RMS / ((last - mean of the deque's window RMSs) * 1000), mu = 10
      'spectrogram' (:55-75)  sqrt(mean of the spectrogram bins above fft_threshold) of the whole deque, mu = 5000
    A NaN energy (e.g. an empty thresholded spectrogram) is "invalid" (-100) and suppresses the step (:100-101, :134)."""

    def __init__(self, initial_angle: float = 0.0, num_win: int = 100, vad_threshold: float = 0.001, method: str = "history",
                 fs: float = 48000.0, fft_threshold: float = 0.00001):
        self.num_win, self.vad_threshold = int(num_win), float(vad_threshold)
        self.method, self.fs, self.fft_threshold = method, float(fs), float(fft_threshold)
        self.mu = 5000.0        # the module-level value; every energy evaluation overwrites it per method (:64, :82)
        self.past_theta = float(initial_angle)
        self.past_energy = INVALID
        self.windows = deque()
        self.n_seen = 0

    def deque_energy(self) -> float:
        if self.method == "spectrogram":
            from scipy import signal
            self.mu = 5000.0
            data = np.concatenate(list(self.windows))
            _, _, spec = signal.spectrogram(data, self.fs, nperseg=1024, noverlap=512, scaling="spectrum")
            sel = spec[spec > self.fft_threshold]
            with np.errstate(invalid="ignore"):
                energy = math.sqrt(np.mean(sel)) if sel.size else float("nan")
        elif self.method == "history":
            self.mu = 10.0
            alpha = 1000
            past = np.array([np.sqrt(np.mean(np.asarray(w) ** 2)) for w in self.windows])
            delta = past[-1] - np.mean(past)
            with np.errstate(divide="ignore", invalid="ignore"):
                energy = float(past[-1] / (delta * alpha))
        else:
            energy = INVALID
        return INVALID if math.isnan(energy) else energy

    def on_windows(self, win, win_ref):
        d = np.asarray(win_ref, dtype=np.float64) - np.asarray(win, dtype=np.float64)
        if self.n_seen < self.num_win:
            self.windows.append(d)
            self.n_seen += 1
            return None
        self.windows.popleft()
        self.windows.append(d)
        if window_rms(d) < self.vad_threshold:
            return None
        if self.past_energy == INVALID:
            self.past_energy = self.deque_energy()
        energy = self.deque_energy()
        if not energy > INVALID:
            return None
        theta = wrap180(self.past_theta + self.mu * (energy - self.past_energy))
        self.past_energy, self.past_theta = energy, theta
        return theta


class SIR2Theta:
    """scripts/SIR2theta.py:9-26: theta = past_theta - mu * (SIR - past_SIR) on every /SIR message; starts at theta = 1.0
    (published once at start-up, :37) with past_SIR = -100; no wrap."""

    def __init__(self, mu: float = 0.01, initial_theta: float = 1.0):
        self.mu, self.past_theta, self.past_sir = float(mu), float(initial_theta), INVALID

    def initial(self) -> float:
        return self.past_theta

    def on_sir(self, sir: float) -> float:
        theta = self.past_theta - self.mu * (float(sir) - self.past_sir)
        self.past_sir, self.past_theta = float(sir), theta
        return theta


class Vad:
    """scripts/vad.py:24-66: the two-state (silence / active) energy detector some launch set-ups run beside the node."""

    def __init__(self, tchange: float = 0.015, tvad: float = 0.02, ehist_len: int = 8, windows_passed_threshold: int = 5):
        self.tchange, self.tvad, self.limit = tchange, tvad, windows_passed_threshold
        self.ehist = np.zeros(ehist_len)
        self.i = 0
        self.enoise = 0.0
        self.passed = 0
        self.silence = False
        self.active = False

    def on_window(self, win) -> bool:
        e = float(np.absolute(np.asarray(win, dtype=np.float64)).mean())
        if (not self.silence) and e > self.enoise + self.tvad:
            self.passed = 0
            self.active = True
        else:
            self.active = False
            self.passed += 1
        mean = float(np.absolute(self.ehist).mean())
        if self.silence and e > mean + self.tchange:
            self.silence = False
            self.enoise = mean
            self.ehist = np.ones(len(self.ehist)) * mean
        elif (not self.silence) and (e < mean - self.tchange or self.passed > self.limit):
            self.passed = 0
            self.silence = True
            self.ehist = np.ones(len(self.ehist)) * self.enoise
        else:
            self.ehist[self.i] = e
            self.i = (self.i + 1) % len(self.ehist)
        return self.active


class JackRef:
    """beamform/src/jack_ref.cpp:19-30 + util.h:353-379 (do_overlap_bymic): the reference channel of the two-topic controllers
    -- one microphone "in the same delayed manner as the rest of the frequency-domain beamformers", i.e. through the WOLA
    framing with NO processing in between: frame t = [hop t-1 | hop t] times the sqrt-Hann window (util.h:235, in double),
    stored as float, times the window again (jack_ref.cpp:26-29, float x double -> float), overlap-added
    (util.h:301-302).  sqrt-Hann^2 at 50 % overlap sums to one, so the output is the input delayed by ONE hop (to float
    rounding): sample-aligned with the beamformer output that energy2theta-diff.py:74 subtracts from it."""

    def __init__(self, hop: int = 512):
        self.H = int(hop)
        n = np.arange(2 * self.H, dtype=np.float64)
        self.win = np.sqrt(0.5 - 0.5 * np.cos(2.0 * np.pi * n / (2 * self.H)))  # util.h:201-211
        self.prev_hop = np.zeros(self.H, np.float32)   # ring pre-filled with one hop of zeros (util.h:272-277)
        self.tail = np.zeros(self.H, np.float32)       # out_buff calloc'ed (util.h:285-286)

    def process_hop(self, s) -> np.ndarray:
        s = np.asarray(s, np.float32)
        frame = np.concatenate([self.prev_hop, s]).astype(np.float64) * self.win   # overlap_and_add_prepare_input
        o = frame.astype(np.float32)                                              # out[j] = real(x[j])
        o = (o.astype(np.float64) * self.win).astype(np.float32)                  # out[j] *= hann_win[j]
        y = self.tail + o[:self.H]                                                # float + float (util.h:301-302)
        self.tail, self.prev_hop = o[self.H:].copy(), s.copy()
        return y


def follow(node, x: np.ndarray, controller, ref_channel: int = 0):
    """The closed loop of the reference, one JACK period at a time: node.process_hop -> controller -> node.set_theta.

    node: anything with process_hop([M, hop]) -> [hop], set_theta(deg) and an attribute H (beamform_amd.capi.Beamformer;
    the tests also drive the oracle node through it).  x: [M, F*hop] float32.  For the two-topic controllers
    (-diff / -spec) the reference channel is microphone `ref_channel` through the jackaudio_ref node (JackRef above): the same
    one-hop WOLA latency as the beamformer output, so the difference the controller forms is sample-aligned.
    Returns (y [F*hop], thetas: list of (hop index, theta) published)."""
    H = node.H
    F = x.shape[1] // H
    ys, published = [], []
    ref = JackRef(H) if hasattr(controller, "on_windows") else None
    for t in range(F):
        seg = np.ascontiguousarray(x[:, t * H:(t + 1) * H])
        y = node.process_hop(seg)
        if isinstance(y, tuple):
            y = y[0]
        ys.append(np.array(y, copy=True))
        if hasattr(controller, "on_windows"):
            theta = controller.on_windows(y, ref.process_hop(seg[ref_channel]))
        else:
            theta = controller.on_window(y)
        if theta is not None:
            node.set_theta(theta)
            published.append((t, theta))
    return np.concatenate(ys), published
