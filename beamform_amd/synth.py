"""Seeded synthetic multichannel scenes (SURVEY.md 8d "common synthetic input").

Used by the parity tests (small scenes) and by bench.py's cpu_baseline sample.
Pure numpy; not on the hot path.
"""
from __future__ import annotations

import numpy as np

from .params import AIRA16_XY

V_SOUND = 343.0


def mic_delays(mics, angle_deg: float) -> np.ndarray:
    """Per-mic delay (s) of a far-field source at `angle_deg`, reference formula (util.h:143-159)."""
    xy = np.asarray(mics, dtype=np.float64)
    dist = np.hypot(xy[:, 0], xy[:, 1])
    ang = np.degrees(np.arctan2(xy[:, 1], xy[:, 0]))
    d = (ang - angle_deg + 180.0) % 360.0 - 180.0
    tau = dist * np.cos(np.radians(d)) / (-V_SOUND)
    tau[0] = 0.0
    return tau


def _bandlimited_noise(rng, T: int, sr: float, lo: float, hi: float, sigma: float) -> np.ndarray:
    spec = np.fft.rfft(rng.standard_normal(T))
    f = np.fft.rfftfreq(T, 1.0 / sr)
    spec[(f < lo) | (f > hi)] = 0.0
    s = np.fft.irfft(spec, T)
    return s * (sigma / (s.std() + 1e-30))


def make_scene(n_mics: int = 8, n_frames: int = 64, hop: int = 512, sr: float = 48000.0, seed: int = 1234,
               mics=None, theta_s: float = 20.0, interferers=(-60.0, 90.0, 150.0), sigma_s: float = 0.2,
               sigma_i: float = 0.1, sigma_n: float = 0.01, silent_frac: float = 0.1) -> np.ndarray:
    """x [n_mics, n_frames*hop] float32: target at theta_s + interferers + sensor noise.

    Sources are band-limited (80 Hz .. 20 kHz) Gaussian noise delayed per mic by
    the reference's own steering delays (exact fractional delays applied in the
    frequency domain over the whole signal), plus white sensor noise; the last
    `silent_frac` of the signal is scaled to ~1e-4 so the magnitude gates of
    mvdr/lcmv/gss/phase see a closed-gate stretch.  Clipped to [-1, 1].
    """
    rng = np.random.default_rng(seed)
    mics = list(AIRA16_XY[:n_mics]) if mics is None else list(mics)
    T = n_frames * hop
    f = np.fft.rfftfreq(T, 1.0 / sr)
    x = np.zeros((n_mics, T))
    for ang, sig in [(theta_s, sigma_s)] + [(a, sigma_i) for a in interferers]:
        s = _bandlimited_noise(rng, T, sr, 80.0, 20000.0, sig)
        S = np.fft.rfft(s)
        tau = mic_delays(mics, ang)
        # mic m hears s(t - tau_m); the reference re-aligns it with conj(w) = exp(+i 2 pi f tau_m)
        x += np.fft.irfft(S[None, :] * np.exp(-2j * np.pi * f[None, :] * tau[:, None]), T)
    x += sigma_n * rng.standard_normal((n_mics, T))
    n_sil = int(T * silent_frac)
    if n_sil:
        x[:, T - n_sil:] *= 1e-4 / max(sigma_s, 1e-12)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def stream_noise(seed: int, n_mics: int, s0: int, s1: int, device="cpu"):
    """Samples [s0, s1) of every channel of ONE global uniform-noise stream in [-0.5, 0.5) -> torch [n_mics, s1-s0] f32.

    Counter-based (a 32-bit integer hash of (seed, mic, sample index)), so any rank can materialise any slice of the
    same stream on its own GPU: the halo a shard re-reads equals the samples its neighbour owns, with no exchange.
    Bench / full-size shard tests only (uniform noise opens every magnitude gate: the worst case for mvdr/lcmv)."""
    import torch
    idx = torch.arange(s0, s1, dtype=torch.int64, device=device)
    out = torch.empty((n_mics, s1 - s0), dtype=torch.float32, device=device)
    for m in range(n_mics):
        h = (idx + ((m + 1) * 0x632BE5AB + (seed + 1) * 0x85157AF5)) & 0xFFFFFFFF
        h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
        h = ((h ^ (h >> 16)) * 0x45D9F3B) & 0xFFFFFFFF
        h = h ^ (h >> 16)
        out[m] = (h >> 8).to(torch.float32) * (1.0 / 16777216.0) - 0.5
    return out
