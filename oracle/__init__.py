"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product package (beamform_amd) never does.

PARITY UNPINNED: see oracle/bf_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

MAX_MICS = 32
MAX_INTERF = 16
ALGO_ID = {"das": 0, "mvdr": 1, "lcmv": 2, "gss": 3, "phase": 4, "phasempf": 5, "mcra": 6, "gsc": 7}


class OrcParams(C.Structure):
    _fields_ = [
        ("algo", C.c_int), ("n_mics", C.c_int), ("hop", C.c_int), ("sample_rate", C.c_double),
        ("mic_x", C.c_double * MAX_MICS), ("mic_y", C.c_double * MAX_MICS), ("theta", C.c_double),
        ("n_interf", C.c_int), ("interf_angle", C.c_double * MAX_INTERF),
        ("past_windows", C.c_int), ("freq_mag_threshold", C.c_double), ("freq_max", C.c_double),
        ("freq_min", C.c_double), ("out_amp", C.c_double), ("mu", C.c_double), ("lambda_", C.c_double),
        ("min_phase", C.c_double), ("mag_mult", C.c_double), ("mag_threshold", C.c_double),
        ("min_mag", C.c_double), ("smooth_size", C.c_int), ("mcra_alphaS", C.c_double), ("mcra_alphaD", C.c_double),
        ("mcra_alphaD2", C.c_double), ("mcra_delta", C.c_double), ("mcra_L", C.c_int), ("mpf_alphaS", C.c_double),
        ("mpf_eta", C.c_double), ("mpf_rev_gamma", C.c_double), ("mpf_rev_delta", C.c_double),
        ("noise_floor", C.c_double), ("out_only_noise", C.c_int), ("out_only_mcra", C.c_int),
        ("gsc_use_vad", C.c_int), ("gsc_vad_threshold", C.c_double), ("gsc_mu0", C.c_double), ("gsc_mu_max", C.c_double),
        ("gsc_filter_size", C.c_int),
    ]


#: defaults of parameters added after the first golden fixtures were written (launch/gsc.launch:6-11)
_LATER_KEYS = dict(gsc_use_vad=0, gsc_vad_threshold=0.1, gsc_mu0=0.0001, gsc_mu_max=0.1, gsc_filter_size=128)


def build(force: bool = False) -> str:
    """Compile liboracle.so with g++ (a few seconds)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH) for f in ("bf_oracle.cpp", "bf_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_theta.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_interference.argtypes = [C.c_void_p, C.c_uint, C.c_double, C.c_double]
        L.orc_process_hop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        for f in ("orc_get_freqs", "orc_get_delays", "orc_get_hann", "orc_get_weights"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.orc_fft.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


def to_struct(p: dict) -> OrcParams:
    s = OrcParams()
    s.algo = ALGO_ID[p["algo"]]
    s.n_mics = p["n_mics"]
    s.hop = p["hop"]
    s.sample_rate = p["sample_rate"]
    for i, (x, y) in enumerate(p["mics"]):
        s.mic_x[i], s.mic_y[i] = x, y
    s.theta = p["theta"]
    s.n_interf = len(p["interf"])
    for i, a in enumerate(p["interf"]):
        s.interf_angle[i] = a
    for k in ("past_windows", "freq_mag_threshold", "freq_max", "freq_min", "out_amp", "mu", "lambda_", "min_phase",
              "mag_mult", "mag_threshold", "min_mag", "smooth_size", "mcra_alphaS", "mcra_alphaD", "mcra_alphaD2",
              "mcra_delta", "mcra_L", "mpf_alphaS", "mpf_eta", "mpf_rev_gamma", "mpf_rev_delta", "noise_floor",
              "out_only_noise", "out_only_mcra", "gsc_use_vad", "gsc_vad_threshold", "gsc_mu0", "gsc_mu_max",
              "gsc_filter_size"):
        setattr(s, k, p[k] if k in p else _LATER_KEYS[k])  # fixtures written before a key existed
    return s


class OracleNode:
    """One reference node (das|mvdr|lcmv|gss|phase|phasempf|mcra|gsc) in double precision on the CPU."""

    def __init__(self, params: dict):
        self.p = params
        self.M = params["n_mics"]
        self.H = params["hop"]
        self.N = 2 * self.H
        self.S = len(params["interf"]) + 1 if params["algo"] in ("lcmv", "gss") else 1
        self._s = to_struct(params)
        self._h = lib().orc_create(C.byref(self._s))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_theta(self, deg: float):
        lib().orc_set_theta(self._h, float(deg))

    def set_interference(self, idx: int, deg: float, threshold: float = 1.0) -> int:
        """interf_theta_roscallback; threshold = interf_angle_threshold (launch files: 1.0).  Returns the interferer count."""
        k = lib().orc_set_interference(self._h, int(idx), float(deg), float(threshold))
        self.S = k + 1
        return k

    def process_hop(self, x: np.ndarray, want_spectrum: bool = False):
        """x: [M, H] float32 -> (out [H] float32, Y [N] complex128 or None)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape == (self.M, self.H)
        out = np.empty(self.H, np.float32)
        Y = np.empty(self.N, np.complex128) if want_spectrum else None
        lib().orc_process_hop(self._h, x.ctypes.data, out.ctypes.data, Y.ctypes.data if want_spectrum else None)
        return out, Y

    def process(self, x: np.ndarray, want_spectrum: bool = False):
        """x: [M, F*H] planar float32 -> (y [F*H] float32, Y [F, N] complex128 or None)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[0] == self.M and x.shape[1] % self.H == 0
        F = x.shape[1] // self.H
        y = np.empty(F * self.H, np.float32)
        Y = np.empty((F, self.N), np.complex128) if want_spectrum else None
        lib().orc_process(self._h, x.ctypes.data, F, y.ctypes.data, Y.ctypes.data if want_spectrum else None)
        return y, Y

    def freqs(self):
        f = np.empty(self.N)
        lib().orc_get_freqs(self._h, f.ctypes.data)
        return f

    def delays(self):
        d = np.empty(self.M)
        lib().orc_get_delays(self._h, d.ctypes.data)
        return d

    def hann(self):
        h = np.empty(self.N)
        lib().orc_get_hann(self._h, h.ctypes.data)
        return h

    def weights(self):
        """[N, M, S] complex128 steering/constraint matrices."""
        w = np.empty((self.N, self.M, self.S), np.complex128)
        lib().orc_get_weights(self._h, w.ctypes.data)
        return w


def fft(x: np.ndarray, sign: int = -1) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    lib().orc_fft(x.ctypes.data, out.ctypes.data, x.size, sign)
    return out
