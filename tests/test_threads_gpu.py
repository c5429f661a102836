"""/theta and /theta_interference arriving on another thread while the callback thread runs (das.cpp:94-99 against
das.cpp:72-92; lcmv.cpp:258-309 against lcmv.cpp:142-162), on the real HIP path.

One thread hammers bf_set_theta (or bf_set_interference) between two values while the main thread runs bf_process_hop
for 2 000+ hops.  An update takes effect at a hop boundary, so every output hop must be
    tail(frame t-1 steered to a) + head(frame t steered to b)      with a, b in {the two values in flight};
a torn table (some microphones from one angle, some from the other) matches none of the four combinations.
The per-frame pieces come from the oracle's spectra under each value (the covariance history of lcmv does not depend on
the steering, so the two oracle runs share it)."""
import threading
import time

import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star tolerance, relative to the RMS level of the stream


def frame_pieces(node, x):
    """Oracle: per-frame windowed output (util.h:244-253) from the node's y_fft -> heads [F, 512], tails [F, 512]."""
    _, Y = node.process(x, want_spectrum=True)
    o = np.real(np.fft.ifft(Y, axis=1)).astype(np.float32)            # Re(IFFT_unnorm)/N, stored as float
    o = (o.astype(np.float64) * node.hann()[None, :]).astype(np.float32)
    if node.p["algo"] in ("mvdr", "lcmv", "gss"):
        o = (o.astype(np.float64) * node.p["out_amp"]).astype(np.float32)
    return o[:, :512], o[:, 512:]


def hammer_and_check(p, x, F, make_nodes, update):
    import oracle  # noqa: F401
    from beamform_amd.capi import Beamformer
    import torch
    assert torch.cuda.is_available()
    pieces = [frame_pieces(n, x) for n in make_nodes()]
    bf = Beamformer(p)
    stop = threading.Event()
    count = [0]

    def ctl():
        i = 0
        while not stop.is_set():
            update(bf, i & 1)
            i += 1
            count[0] = i
            if i % 16 == 0:
                time.sleep(0.0002)

    th = threading.Thread(target=ctl)
    th.start()
    try:
        y = np.stack([bf.process_hop(x[:, t * 512:(t + 1) * 512]) for t in range(F)])
    finally:
        stop.set()
        th.join()
    assert count[0] > 200, "the control thread hardly ran"
    allh = np.concatenate([h for h, _ in pieces]).astype(np.float64)
    rms = float(np.sqrt(np.mean(allh[np.isfinite(allh)] ** 2)))       # mvdr/lcmv: frame 0 of a cold node is NaN
    used = set()
    for t in range(1, F):
        best, arg = None, None
        for a in range(2):
            for b in range(2):
                cand = pieces[a][1][t - 1] + pieces[b][0][t]
                if not np.isfinite(cand).all():
                    continue
                e = float(np.sqrt(np.mean((y[t].astype(np.float64) - cand) ** 2))) / rms
                if best is None or e < best:
                    best, arg = e, (a, b)
        if best is None:
            continue                                        # reference output itself is NaN (mvdr/lcmv frame 0)
        assert np.isfinite(y[t]).all() and best < TOL, (t, best)
        used.add(arg)
    return used, count[0]


def test_theta_hammered_while_das_callbacks_run():
    import oracle
    M, F = 4, 2200
    A, B = 20.0, -75.0
    p = make_params("das", n_mics=M, theta=A)
    x = make_scene(M, F, seed=77, silent_frac=0.0)
    used, n = hammer_and_check(
        p, x, F, lambda: [oracle.OracleNode(make_params("das", n_mics=M, theta=a)) for a in (A, B)],
        lambda bf, k: bf.set_theta(B if k else A))
    assert len(used) >= 2, (used, n)                        # both angles were really seen by the callback thread


def test_interferer_hammered_while_lcmv_callbacks_run():
    import oracle
    M, F = 4, 2100
    I1, I2 = -60.0, 100.0
    p = make_params("lcmv", n_mics=M, theta=20.0, interf=(I1,))
    x = make_scene(M, F, seed=78, silent_frac=0.0)
    used, n = hammer_and_check(
        p, x, F, lambda: [oracle.OracleNode(make_params("lcmv", n_mics=M, theta=20.0, interf=(i,))) for i in (I1, I2)],
        lambda bf, k: bf.set_interference(1, I2 if k else I1))
    assert len(used) >= 2, (used, n)
