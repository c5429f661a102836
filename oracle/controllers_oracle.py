"""Restatement of the reference's /theta controller callbacks (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Each function follows one script of /root/reference/beamform/scripts/ callback by callback, with the script's module-level
globals kept in a dict `g`:

  ref_energy2theta        energy2theta.py:23-27 (get_energy_from_list), :29-60 (get_energy_from_deque), :62-101 (energycallback)
  ref_energy2theta_diff   energy2theta-diff.py:31-62, :72-103
  ref_energy2theta_spec   energy2theta-spec.py:37-103 (both energy_calc_method values: 'history' and 'spectrogram'), :105-150
  ref_sir2theta           SIR2theta.py:9-26 (+ the start-up publish, :37)

A callback receives what rospy hands it: the float32[] field of a JackAudio message as a tuple of Python floats.  Every
function returns the list of (message index, published theta).

PARITY UNPINNED: the scripts need rospy, message_filters, jack_msgs and (the -spec one) a GTK matplotlib backend, none of which
is in the image, so they cannot be run here; these are this repo's reading of them.  The fixtures
tests/golden/controllers_*.npz (tests/golden/make_controllers_golden.py) pin this reading against accidental change; they
are not reference outputs.  Only tests/ may import this module; the product's controllers are beamform_amd/controllers.py.
"""
import math
from collections import deque

import numpy as np


def ref_energy_from_list(data_list):                              # energy2theta.py:23-27
    sq = [i ** 2 for i in data_list]
    return math.sqrt(sum(sq) / len(sq))


def _wrap(theta):                                                 # energy2theta.py:85-88 (the same four lines in every script)
    if theta > 180:
        theta = theta - 360
    elif theta < -180:
        theta = theta + 360
    return theta


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
                theta = _wrap(g["past_theta"] + mu * (energy - g["past_energy"]))
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
            theta = _wrap(g["past_theta"] - mu * (energy - g["past_energy"]))
            out.append((k, theta))
            g["past_energy"], g["past_theta"] = energy, theta
    return out


def ref_energy2theta_spec(pairs, initial_angle, method="history", num_win=100, vad_threshold=0.001, fs=48000, fft_threshold=0.00001):
    """energy_calc_method = `method` (energy2theta-spec.py:19; the script ships with 'history')."""
    g = dict(num_win_i=0, past_energy=-100.0, past_theta=initial_angle, windows=deque([]), mu=5000)
    out = []

    def e():                                                      # energy2theta-spec.py:37-103
        if method == "spectrogram":                               # :55-75
            from scipy import signal
            g["mu"] = 5000
            data_np = np.array([item for sub in list(g["windows"]) for item in sub])
            _, _, spec_data = signal.spectrogram(data_np, fs, nperseg=1024, noverlap=512, scaling="spectrum")
            spec_data_filt = spec_data[spec_data > fft_threshold]
            with np.errstate(invalid="ignore"), np.testing.suppress_warnings() as sup:
                sup.filter(RuntimeWarning)
                energy = math.sqrt(np.mean(spec_data_filt)) if spec_data_filt.size else float("nan")  # np.mean([]) is nan there too
        elif method == "history":                                 # :77-94
            g["mu"] = 10
            alpha = 1000
            past_values = np.array([np.sqrt(np.mean(np.array(w) ** 2)) for w in list(g["windows"])])
            delta = past_values[-1] - np.mean(past_values)
            with np.errstate(divide="ignore", invalid="ignore"):
                energy = past_values[-1] / (delta * alpha)
        else:
            energy = -100.0
        return -100.0 if math.isnan(energy) else energy           # :100-101

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
                    theta = _wrap(g["past_theta"] + g["mu"] * (energy - g["past_energy"]))
                    out.append((k, theta))
                    g["past_energy"], g["past_theta"] = energy, theta
    return out


def ref_sir2theta(sirs, mu=0.01):                                 # SIR2theta.py:9-26; theta = 1.0 is published once at start-up (:37)
    past_sir, past_theta, out = -100.0, 1.0, []
    for k, sir in enumerate(sirs):
        theta = past_theta - mu * (sir - past_sir)
        out.append((k, theta))
        past_sir, past_theta = sir, theta
    return out
