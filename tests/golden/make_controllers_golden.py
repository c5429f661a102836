#!/usr/bin/env python3
"""Regenerates tests/golden/controllers_loop.npz: a window stream in, the published theta sequences out.

PARITY UNPINNED (see oracle/controllers_oracle.py): the reference's controller scripts need rospy / message_filters / jack_msgs,
which the image lacks, so the theta sequences come from this repo's restatement of their callbacks driven by the CPU oracle's
das node.  The file pins that restatement and travels to the GPU box as data; it is NOT a reference output.

Content: scene parameters (json: regenerate the input with beamform_amd.synth.make_scene), win [F, 512] f32 = the das node's output
windows (the `jackaudio` topic), ref [F, 512] f32 = microphone 0 (the `jackaudio_ref` topic as the -diff / -spec tests feed it), and
per controller the message indices k_<name> and published angles theta_<name> (float64, exactly as computed).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import controllers_oracle as co  # noqa: E402
from beamform_amd.params import make_params  # noqa: E402
from beamform_amd.synth import make_scene  # noqa: E402

SCENE = dict(n_mics=4, n_frames=170, seed=41, theta_s=20.0, silent_frac=0.12, node_theta=-15.0)


def sequences(win, ref, initial_angle):
    as_msg = lambda a: tuple(float(v) for v in a)   # rospy hands float32[] fields over as tuples of Python floats
    ys, rs = [as_msg(w) for w in win], [as_msg(r) for r in ref]
    pairs = list(zip(ys, rs))
    return {
        "energy": co.ref_energy2theta(ys, initial_angle),
        "diff": co.ref_energy2theta_diff(pairs, initial_angle),
        "spec_history": co.ref_energy2theta_spec(pairs, initial_angle, method="history"),
        # 30 windows instead of the script's 100 so that the 170-window stream leaves steps to compare (num_win is its module constant)
        "spec_spectrogram": co.ref_energy2theta_spec(pairs, initial_angle, method="spectrogram", num_win=30),
    }


def main():
    sc = SCENE
    p = make_params("das", n_mics=sc["n_mics"], theta=sc["node_theta"])
    x = make_scene(sc["n_mics"], sc["n_frames"], seed=sc["seed"], theta_s=sc["theta_s"], silent_frac=sc["silent_frac"])
    y, _ = oracle.OracleNode(p).process(x)
    F = sc["n_frames"]
    win = y.reshape(F, 512).astype(np.float32)
    ref = x[0].reshape(F, 512).astype(np.float32)
    out = dict(scene=json.dumps(sc), win=win, ref=ref)
    for name, seq in sequences(win, ref, sc["node_theta"]).items():
        assert len(seq) > 20, (name, len(seq))
        out["k_" + name] = np.array([k for k, _ in seq], np.int32)
        out["theta_" + name] = np.array([t for _, t in seq], np.float64)
        print(name, len(seq), "steps, theta", out["theta_" + name][0], "...", out["theta_" + name][-1])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "controllers_loop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
