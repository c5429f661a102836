#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

PARITY UNPINNED: the reference has no tests or fixtures and cannot be built here, so these vectors
are produced by this repo's CPU oracle (oracle/bf_oracle.cpp) and accepted only after the independent
numpy restatement (oracle/np_oracle.py) reproduces them (max per-frame relative L2 on the spectrum
< 1e-10, time signal bit-identical).  They pin the oracle against accidental change and travel to the
GPU box as data; they are NOT reference outputs.

Each file: params (json), x [M, F*512] f32, y [F*512] f32, Y [F, 1024] c128.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import np_oracle  # noqa: E402
from beamform_amd.params import make_params  # noqa: E402
from beamform_amd.synth import make_scene  # noqa: E402

CASES = [
    # name, algo, M, interferers, frames, theta  (sizes keep every file < 1 MB after float16-free compression)
    ("das4", "das", 4, (), 12, 0.0),        # BASELINE config 1 (4-mic das, streaming plumbing)
    ("das8", "das", 8, (), 10, 20.0),       # config 2 shape
    ("mvdr8", "mvdr", 8, (), 16, 20.0),     # config 3 shape
    ("phasempf8", "phasempf", 8, (), 16, 20.0),  # config 4 shape
    ("lcmv16", "lcmv", 16, (-60.0, 90.0, 150.0), 14, 20.0),  # config 5 shape
    ("gss8", "gss", 8, (-60.0, 90.0), 12, 20.0),
    ("phase8", "phase", 8, (), 10, 20.0),
    ("mcra2", "mcra", 2, (), 24, 0.0, dict(mcra_L=8)),
    ("gsc4", "gsc", 4, (), 8, 20.0),  # SURVEY 8(f) row 1; time-domain node: Y is all zeros by definition  # SURVEY 8(f) row 2; short L puts the minima reset inside the run
]


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])  # optional: names to (re)generate; default all
    for name, algo, M, interf, F, theta, *over in CASES:
        if only and name not in only:
            continue
        p = make_params(algo, n_mics=M, interf=interf, theta=theta, **(over[0] if over else {}))
        x = make_scene(M, F, seed=sum(map(ord, name)))
        y, Y = oracle.OracleNode(p).process(x, want_spectrum=True)
        y2, Y2 = np_oracle.process(p, x)
        fin = np.isfinite(Y).all(axis=1)
        assert (fin == np.isfinite(Y2).all(axis=1)).all(), name
        worst = max(np.linalg.norm(Y[t] - Y2[t]) / (np.linalg.norm(Y2[t]) or 1.0) for t in range(F) if fin[t])
        assert worst < 1e-10, (name, worst)
        ok = np.isfinite(y)
        if algo == "gsc":  # the two FFTs differ by ~1e-16 absolute on near-zero samples
            assert np.abs(y[ok] - y2[ok]).max() < 1e-6 * np.abs(y[ok]).max(), name
        else:
            assert np.array_equal(y[ok], y2[ok]), name
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), params=json.dumps(p), x=x, y=y, Y=Y)
        print(f"{name}: {F} frames, cross-check {worst:.1e}, finite frames {int(fin.sum())}/{F}")


if __name__ == "__main__":
    main()
