"""precision of a truncated-double (48-bit) packed spectrum in the mvdr solve (numpy, batched)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from beamform_amd.synth import make_scene
from beamform_amd.params import make_params
from oracle import np_oracle as npo

def trunc(a, bits_drop=16, rnd=True):
    u = a.view(np.uint64).copy()
    if rnd: u = u + np.uint64(1 << (bits_drop - 1))
    u &= ~np.uint64((1 << bits_drop) - 1)
    return u.view(np.float64)

def mvdr(p, X, w):
    F, M, N = X.shape; P = p["past_windows"]
    f = np.abs(npo.freq_vector(N, p["sample_rate"]))
    bins = np.where((f >= p["freq_min"]) & (f <= p["freq_max"]))[0]
    bins = bins[bins < N // 2]
    Xb = X[:, :, bins]                                  # F M B
    Y = np.zeros((F, len(bins)), complex); conds = []
    white = np.ones((M, M)) + 0.001 * np.eye(M)
    for t in range(P, F):
        H = Xb[t - P:t]                                 # P M B
        R = np.einsum("pmb,pnb->bmn", H, H.conj()) * white
        a = w[:, bins].T                                # B M
        z = np.linalg.solve(R, a[..., None])[..., 0]
        wopt = z / np.einsum("bm,bm->b", a.conj(), z)[:, None]
        Y[t] = np.einsum("bm,mb->b", wopt.conj(), Xb[t])
        if t % 16 == 0: conds.append(np.linalg.cond(R))
    return Y, np.concatenate(conds)

for M in (8, 16):
    for kind in ("scene", "noise"):
        p = make_params("mvdr", n_mics=M, theta=20.0)
        F = 60
        x = make_scene(M, F, seed=3, silent_frac=0.0) if kind == "scene" else (np.random.default_rng(1).random((M, F * 512), dtype=np.float32) - 0.5)
        X = npo.stft(p, x)
        w = npo.steering(p, 20.0)
        Y0, conds = mvdr(p, X, w)
        for drop in (16, 20, 24, 29):
            # pack pairs Z = Xa + i Xb, truncate components, unpack
            Xa, Xb_ = X[:, 0::2], X[:, 1::2]
            Z = Xa + 1j * Xb_
            Zt = trunc(Z.real.copy(), drop) + 1j * trunc(Z.imag.copy(), drop)
            Zm = np.conj(np.roll(Zt[..., ::-1], 1, axis=-1))  # conj Z[N-k]
            Xq = np.empty_like(X); Xq[:, 0::2] = (Zt + Zm) / 2; Xq[:, 1::2] = (Zt - Zm) / 2j
            Y1, _ = mvdr(p, Xq, w)
            e = np.linalg.norm(Y1[10:] - Y0[10:], axis=1) / np.linalg.norm(Y0[10:], axis=1)
            print(f"M={M} {kind}: cond med {np.median(conds):.2e} max {conds.max():.2e}  drop {drop} bits (mant {52-drop}): rel L2 per frame max {e.max():.2e} med {np.median(e):.2e}")
