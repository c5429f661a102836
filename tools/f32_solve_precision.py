"""mvdr with the Cholesky solve in complex64 (R accumulated in double), vs all-double: relative L2 of the solved spectrum per frame."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from beamform_amd.synth import make_scene
from beamform_amd.params import make_params
from oracle import np_oracle as npo

def chol_solve(R, a, x, dt):
    # R = L L^H ; u = L^-1 a ; v = L^-1 x ; y = u^H v / u^H u   (batched over leading dim), arithmetic in dtype dt
    R = R.astype(dt); a = a.astype(dt); x = x.astype(dt)
    B, M, _ = R.shape
    A = R.copy(); ua = a.copy(); ux = x.copy()
    rt = np.float32 if dt == np.complex64 else np.float64
    for j in range(M):
        inv = (1.0 / np.sqrt(A[:, j, j].real.astype(rt))).astype(rt)
        ua[:, j] = ua[:, j] * inv; ux[:, j] = ux[:, j] * inv
        for i in range(j + 1, M):
            Lij = (A[:, i, j] * inv).astype(dt)
            A[:, i, j] = Lij
            ua[:, i] = (ua[:, i] - Lij * ua[:, j]).astype(dt)
            ux[:, i] = (ux[:, i] - Lij * ux[:, j]).astype(dt)
        for c in range(j + 1, M):
            Lc = A[:, c, j]
            for i in range(c, M):
                A[:, i, c] = (A[:, i, c] - A[:, i, j] * np.conj(Lc)).astype(dt)
    num = np.sum(ux * np.conj(ua), axis=1); den = np.sum(np.abs(ua) ** 2, axis=1)
    return (num / den).astype(np.complex128)

def run(M, kind, seed):
    p = make_params("mvdr", n_mics=M, theta=20.0); P = p["past_windows"]
    F = 48
    x = make_scene(M, F, seed=seed, silent_frac=0.0) if kind == "scene" else (np.random.default_rng(seed).random((M, F * 512), dtype=np.float32) - 0.5)
    X = npo.stft(p, x); w = npo.steering(p, 20.0)
    N = X.shape[2]; f = np.abs(npo.freq_vector(N, p["sample_rate"]))
    bins = np.where((f >= p["freq_min"]) & (f <= p["freq_max"]))[0]; bins = bins[bins < N // 2]
    Xb = X[:, :, bins]; a = w[:, bins].T
    white = np.ones((M, M)) + 0.001 * np.eye(M)
    errs = []
    for t in range(P, F):
        H = Xb[t - P:t]
        R = np.einsum("pmb,pnb->bmn", H, H.conj()) * white
        xt = Xb[t].T
        y64 = chol_solve(R, a, xt, np.complex128)
        y32 = chol_solve(R, a, xt, np.complex64)
        errs.append(np.linalg.norm(y32 - y64) / np.linalg.norm(y64))
    return np.array(errs)

for M in (8, 6, 4):
    for kind in ("scene", "noise"):
        e = np.concatenate([run(M, kind, s) for s in (3, 4)])
        print(f"M={M} {kind}: fp32 solve rel L2 per frame: median {np.median(e):.2e} max {e.max():.2e}")
