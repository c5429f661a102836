"""Independent numpy restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).

Purpose: a second opinion on oracle/bf_oracle.cpp, written batch-wise on top of
numpy.fft (pocketfft) and numpy.linalg.inv (LAPACK getrf/getri = LU with partial
pivoting, the same factorisation family Eigen's inverse() uses), so an index or
convention slip in the C++ restatement shows up as a disagreement.  It follows
the same reference lines (cited per function) but shares no code with the C++.

PARITY UNPINNED: the reference has no tests or golden vectors and cannot be
built in this image; two independent restatements agreeing is the strongest pin
available here.
"""
from __future__ import annotations

import numpy as np

V_SOUND = 343.0


def hann_sqrt(N: int) -> np.ndarray:
    """util.h:201-211: periodic sqrt-Hann."""
    n = np.arange(N, dtype=np.float64)
    return np.sqrt(0.5 - 0.5 * np.cos(2 * np.pi * n / N))


def freq_vector(N: int, sr: float) -> np.ndarray:
    """util.h:190-199 with Q1: f[N/2-1] overwritten by sr/2, f[N/2] := 0."""
    f = np.zeros(N)
    k = np.arange(1, N // 2)
    f[k] = k / N * sr
    f[N - k] = -(k / N) * sr
    f[N // 2 - 1] = sr / 2
    return f


def delays_for(mics, angle_deg: float) -> np.ndarray:
    """util.h:82-92 (dist/angle from raw xy) + util.h:136-161."""
    xy = np.asarray(mics, dtype=np.float64)
    dist = np.sqrt(xy[:, 0] ** 2 + xy[:, 1] ** 2)
    ang = np.degrees(np.arctan2(xy[:, 1], xy[:, 0]))
    d = ang - angle_deg
    d = np.where(d > 180, d - 360, np.where(d < -180, d + 360, d))
    tau = dist * np.cos(np.radians(d)) / (-V_SOUND)
    tau[0] = 0.0
    return tau


def steering(p: dict, angle_deg: float) -> np.ndarray:
    """das.cpp:27-45: w[m, j] = exp(-i 2 pi f_j tau_m); row 0 = 1.  -> [M, N]"""
    N = 2 * p["hop"]
    f = freq_vector(N, p["sample_rate"])
    tau = delays_for(p["mics"], angle_deg)
    w = np.exp(-1j * 2 * np.pi * f[None, :] * tau[:, None])
    w[0, :] = 1.0
    return w


def stft(p: dict, x: np.ndarray) -> np.ndarray:
    """util.h:217-242,272-277 + das.cpp:51-57 -> X [F, M, N] complex128."""
    M, H = p["n_mics"], p["hop"]
    N = 2 * H
    F = x.shape[1] // H
    xp = np.concatenate([np.zeros((M, H), np.float32), x.astype(np.float32)], axis=1).astype(np.float64)
    h = hann_sqrt(N)
    idx = (np.arange(F) * H)[:, None] + np.arange(N)[None, :]
    frames = xp[:, idx] * h  # [M, F, N]
    return np.fft.fft(frames, axis=-1).transpose(1, 0, 2)


def istft_ola(p: dict, Y: np.ndarray, post_amp=None) -> np.ndarray:
    """fftw backward + util.h:244-253 + util.h:301-302 (float32 rounding as the reference stores it)."""
    H = p["hop"]
    N = 2 * H
    F = Y.shape[0]
    h = hann_sqrt(N)
    yt = np.fft.ifft(Y, axis=-1) * N
    o = (yt.real / N).astype(np.float32)
    o = (o.astype(np.float64) * h).astype(np.float32)
    if post_amp is not None:
        o = (o.astype(np.float64) * post_amp).astype(np.float32)
    prev = np.concatenate([np.zeros((1, N), np.float32), o[:-1]], axis=0)
    out = prev[:, H:] + o[:, :H]
    return out.reshape(F * H).astype(np.float32)


def _pair_phase_mean(phases: np.ndarray) -> np.ndarray:
    """phase.cpp:53-68: mean over pairs m<m' of the wrapped |p_m - p_m'|.  phases [..., M]"""
    M = phases.shape[-1]
    tot = np.zeros(phases.shape[:-1])
    cnt = 0
    for a in range(M - 1):
        for b in range(a + 1, M):
            d = np.abs(phases[..., a] - phases[..., b])
            d = np.where(d > np.pi, 2 * np.pi - d, d)
            tot = tot + d
            cnt += 1
    with np.errstate(invalid="ignore", divide="ignore"):
        return tot / cnt if cnt else tot / 0.0


def das_bins(p, X, w):
    """das.cpp:60-63."""
    return np.einsum("mj,fmj->fj", np.conj(w), X) / p["n_mics"]


def mvdr_lcmv_bins(p, X, C, lcmv: bool):
    """mvdr.cpp:76-105 / lcmv.cpp:102-130.  C: [N, M, S] constraint matrices."""
    F, M, N = X.shape
    P = p["past_windows"]
    f = np.abs(freq_vector(N, p["sample_rate"]))
    inband = (f >= p["freq_min"]) & (f <= p["freq_max"])
    Y = np.zeros((F, N), np.complex128)
    white = np.ones((M, M)) + 0.001 * np.eye(M)
    hist = np.zeros((N, M, P), np.complex128)
    for t in range(F):
        x = X[t]  # [M, N]
        mag = np.abs(x).sum(axis=0) / (M * N)
        for j in range(N):
            if j == 0 and not lcmv:
                Y[t, 0] = x[0, 0]
                continue
            if not inband[j]:
                continue
            if mag[j] > p["freq_mag_threshold"]:
                R = (hist[j] @ hist[j].conj().T) * white
                with np.errstate(all="ignore"):
                    try:
                        Ri = np.linalg.inv(R)
                    except np.linalg.LinAlgError:
                        Ri = np.full((M, M), np.nan + 0j)
                    Cj = C[j]
                    if not lcmv:
                        a = Cj[:, 0]
                        wopt = (Ri @ a) / (a.conj() @ Ri @ a)
                    else:
                        G = Cj.conj().T @ Ri @ Cj
                        try:
                            Gi = np.linalg.inv(G)
                        except np.linalg.LinAlgError:
                            Gi = np.full(G.shape, np.nan + 0j)
                        wopt = ((Ri @ Cj) @ Gi)[:, 0]
                    Y[t, j] = wopt.conj() @ x[:, j]
            else:
                Y[t, j] = 0.01 * x[0, j]
            hist[j, :, :-1] = hist[j, :, 1:]
            hist[j, :, -1] = x[:, j]
    return Y


def gss_bins(p, X, C):
    """gss.cpp:110-146.  C: [N, M, S]."""
    F, M, N = X.shape
    S = C.shape[2]
    f = np.abs(freq_vector(N, p["sample_rate"]))
    inband = (f >= p["freq_min"]) & (f <= p["freq_max"])
    W = np.conj(np.transpose(C, (0, 2, 1))).copy()  # [N, S, M] = C^H
    Ch = W.copy()
    Y = np.zeros((F, N), np.complex128)
    c2 = 2 * (1 // S)  # integer division, Q13
    mu, lam = p["mu"], p["lambda_"]
    for t in range(F):
        x = X[t]
        mag = np.abs(x).sum(axis=0) / (M * N)
        for j in np.nonzero(inband)[0]:
            if mag[j] > p["freq_mag_threshold"]:
                xj = x[:, j]
                y = W[j] @ xj
                Y[t, j] = y[0]
                E = np.outer(y, y.conj())
                np.fill_diagonal(E, 0)
                alpha = (np.abs(xj) ** 2).sum() ** 2
                dj1 = 4 * S * (1 / alpha) * np.outer(E @ y, xj.conj())
                dj2 = c2 * ((W[j] @ C[j]) - np.eye(S)) @ Ch[j]
                W[j] = (1 - lam * mu) * W[j] - mu * (dj1 + dj2)
            else:
                Y[t, j] = 0.01 * x[0, j]
    return Y


def phase_bins(p, X, w):
    """phase.cpp:87-127."""
    F, M, N = X.shape
    mag = np.abs(X).sum(axis=1) / M  # [F, N]
    pha = np.angle(X[:, 0, :])
    aligned = np.angle(np.conj(w)[None] * X)  # [F, M, N]
    dmean = _pair_phase_mean(np.moveaxis(aligned, 1, -1))
    thr = p["min_phase"] * np.pi / 180
    keep = (mag / N > p["mag_threshold"]) & (dmean < thr)
    m2 = np.where(keep, mag, mag * p["mag_mult"])
    Y = m2 * np.cos(pha) + 1j * (m2 * np.sin(pha))
    Y[:, 0] = X[:, 0, 0]
    return Y


def phasempf_bins(p, X, w):
    """phasempf.cpp:210-295 incl. mcra() :140-191.  Returns Y [F, N]."""
    F, M, N = X.shape
    thr = p["min_phase"] * np.pi / 180
    aS, aD, aD2, delta, L = p["mcra_alphaS"], p["mcra_alphaD"], p["mcra_alphaD2"], p["mcra_delta"], p["mcra_L"]
    Sprev = np.zeros(N); Stmp = np.zeros(N); Smin = np.zeros(N); lam = np.zeros(N)
    Z = np.zeros(N); rev0 = np.zeros(N); rev1 = np.zeros(N)
    cL, firstL = 0, True
    taps = np.ones(N); taps[1] = 0.75; taps[N - 1] = 0.75
    Y = np.zeros((F, N), np.complex128)
    for t in range(F):
        x = X[t]
        mag = np.abs(x).sum(axis=0) / M
        pha = np.angle(x[0])
        aligned = np.angle(np.conj(w) * x)  # [M, N]
        dmean = _pair_phase_mean(aligned.T)
        is_soi = dmean < thr
        msoi = np.where(is_soi, mag, mag * p["min_mag"])
        mint = np.where(is_soi, mag * p["min_mag"], mag)
        soi = msoi * np.cos(pha) + 1j * (msoi * np.sin(pha))
        inter = mint * np.cos(pha) + 1j * (mint * np.sin(pha))
        soi[0] = x[0, 0]; inter[0] = x[0, 0]
        soi2 = np.abs(soi) ** 2; int2 = np.abs(inter) ** 2
        soi2[0] = 0.0; int2[0] = 0.0  # defined value for the reference's unwritten index 0
        Sf = taps * soi2
        Sf[0] = np.abs(soi[0])
        S = aS * Sprev + (1 - aS) * Sf
        if cL > L:
            Smin = np.minimum(Stmp, S); Stmp = S.copy(); cL = 1; firstL = False
        else:
            Smin = np.minimum(Smin, S); Stmp = np.minimum(Stmp, S); cL += 1
        cond = (S < Smin * delta) | (lam > soi2)
        if firstL:
            cond = np.ones(N, bool)
        if firstL and (1.0 / cL) > aD:
            new = (1.0 / cL) * lam + (1.0 - 1.0 / cL) * soi2
        else:
            new = aD2 * lam + (1.0 - aD) * soi2
        lam = np.where(cond, new, lam)
        Sprev = S
        Z = p["mpf_alphaS"] * Z + (1 - p["mpf_alphaS"]) * int2
        k = 1 - p["mpf_rev_gamma"] / p["mpf_rev_delta"]
        rev0 = p["mpf_rev_gamma"] * rev0 + k * soi2
        rev1 = p["mpf_rev_gamma"] * rev1 + k * int2
        Lam = np.sqrt(lam + p["mpf_eta"] * Z + rev0 + rev1)
        ph = np.angle(soi)
        if p["out_only_noise"]:
            g = Lam * p["out_amp"]
        else:
            g = (np.abs(soi) - (np.sqrt(lam) if p["out_only_mcra"] else Lam)) * p["out_amp"]
            g = np.where(g < 0, p["noise_floor"], g)
        Yt = g * np.cos(ph) + 1j * (g * np.sin(ph))
        Yt[0] = 0.0
        Y[t] = Yt
    return Y


def mcra_node_bins(p, X):
    """mcra.cpp:64-155 (channel 0 only).  Returns Y [F, N]."""
    F, M, N = X.shape
    aS, aD, aD2, delta, L = p["mcra_alphaS"], p["mcra_alphaD"], p["mcra_alphaD2"], p["mcra_delta"], p["mcra_L"]
    Sprev = np.zeros(N); Stmp = np.zeros(N); Smin = np.zeros(N); lam = np.zeros(N)
    cL, firstL = 0, True
    Y = np.zeros((F, N), np.complex128)
    for t in range(F):
        x = X[t, 0]
        x2 = x.real ** 2 + x.imag ** 2
        lo = np.concatenate([[0.0], x2[:-1]]); lo[1] = 0.0   # bin 0 is excluded from the smoothing window
        hi = np.concatenate([x2[1:], [0.0]])                 # bin N does not exist
        Sf = (0.25 * lo + 0.5 * x2) + 0.25 * hi
        Sf[0] = np.abs(x[0])
        S = aS * Sprev + (1 - aS) * Sf
        if cL > L:
            Smin = np.minimum(Stmp, S); Stmp = S.copy(); cL = 1; firstL = False
        else:
            Smin = np.minimum(Smin, S); Stmp = np.minimum(Stmp, S); cL += 1
        cond = np.ones(N, bool) if firstL else ((S < Smin * delta) | (lam > x2))
        if firstL and (1.0 / cL) > aD:
            new = (1.0 / cL) * lam + (1.0 - 1.0 / cL) * x2
        else:
            new = aD2 * lam + (1.0 - aD) * x2
        lam = np.where(cond, new, lam)
        Sprev = S
        ph = np.angle(x)
        if p["out_only_noise"]:
            g = np.sqrt(lam) * p["out_amp"]
        else:
            g = np.maximum((np.abs(x) - np.sqrt(lam)) * p["out_amp"], 0.0)
        Yt = g * np.cos(ph) + 1j * (g * np.sin(ph))
        Yt[0] = 0.0  # never written by the node
        Y[t] = Yt
    return Y


def gsc_process(p: dict, x: np.ndarray) -> np.ndarray:
    """gsc.cpp:54-75 (per-mic alignment through the STFT) + gsc.cpp:120-181 (float32 NLMS, sample by sample).
    Sequential float32 sums are np.cumsum(...)[-1] (cumsum accumulates left to right in the array dtype)."""
    f32 = np.float32
    M, fs = p["n_mics"], p["gsc_filter_size"]
    X = stft(p, x)                      # [F, M, N]
    w = steering(p, p["theta"])         # [M, N]
    a = np.stack([istft_ola(p, np.conj(w[m]) * X[:, m, :]) for m in range(M)])  # [M, T] float32
    T = a.shape[1]
    bm = np.zeros((max(M - 1, 0), fs), f32); flt = np.zeros_like(bm); lo = np.zeros(fs, f32)
    out = np.zeros(T, f32)
    mu0, mu_max = p["gsc_mu0"], p["gsc_mu_max"]
    with np.errstate(all="ignore"):
        for j in range(T):
            das = f32(0.0)
            for m in range(M):
                das = f32(das + a[m, j])
            o = f32(das / f32(M))
            if M > 1:
                bm[:, :-1] = bm[:, 1:]
                bm[:, -1] = a[1:, j] - a[:-1, j]
                bo = np.cumsum(flt * bm, axis=1, dtype=f32)[:, -1]
                for i in range(M - 1):
                    o = f32(o - bo[i])
            lo[:-1] = lo[1:]; lo[-1] = o
            lop = f32(np.sqrt(f32(np.cumsum(lo * lo, dtype=f32)[-1] / f32(fs))))
            out[j] = o
            if lop < p["gsc_vad_threshold"] or not p["gsc_use_vad"]:
                for i in range(M - 1):
                    bp = f32(np.sqrt(f32(np.cumsum(bm[i] * bm[i], dtype=f32)[-1] / f32(fs))))
                    if np.float64(mu0) * np.float64(bp) / np.float64(lop) < mu_max:
                        mu = f32(np.float64(mu0) / np.float64(lop))
                    else:
                        mu = f32(np.float64(mu0) / np.float64(bp))
                    if not np.isfinite(mu):
                        mu = f32(0.0)
                    upd = flt[i] + f32(mu * o) * bm[i]
                    flt[i] = np.where(np.isnan(upd), f32(0.0), upd)
    return out


def constraint_matrices(p: dict, theta: float) -> np.ndarray:
    """lcmv.cpp:44-86: C_j = [steer, interferer_1..K] -> [N, M, S]."""
    cols = [steering(p, theta)] + [steering(p, a) for a in p["interf"]]
    return np.stack(cols, axis=-1).transpose(1, 0, 2)


def process(p: dict, x: np.ndarray):
    """Whole stream from cold start: x [M, F*H] float32 -> (y [F*H] float32, Y [F, N] complex128)."""
    algo = p["algo"]
    if algo == "gsc":  # time-domain node: no single output spectrum
        y = gsc_process(p, x)
        return y, np.zeros((len(y) // p["hop"], 2 * p["hop"]), np.complex128)
    X = stft(p, x)
    w = steering(p, p["theta"])
    post = None
    if algo == "das":
        Y = das_bins(p, X, w)
    elif algo in ("mvdr", "lcmv"):
        C = constraint_matrices(p, p["theta"]) if algo == "lcmv" else w.T[:, :, None]
        Y = mvdr_lcmv_bins(p, X, C, algo == "lcmv")
        post = p["out_amp"]
    elif algo == "gss":
        Y = gss_bins(p, X, constraint_matrices(p, p["theta"]))
        post = p["out_amp"]
    elif algo == "phase":
        Y = phase_bins(p, X, w)
    elif algo == "phasempf":
        Y = phasempf_bins(p, X, w)
    elif algo == "mcra":
        Y = mcra_node_bins(p, X)
    else:
        raise ValueError(algo)
    y = istft_ola(p, Y, post)
    if algo == "phasempf":  # phasempf.cpp:331-334 moving average over output samples
        sz = p["smooth_size"]
        buf = np.concatenate([np.zeros(sz - 1), y.astype(np.float64)])
        acc = np.zeros(len(y))
        for i in range(sz):  # same summation order as get_mean (oldest first)
            acc = acc + buf[i:i + len(y)]
        y = (acc / sz).astype(np.float32)
    return y, Y
