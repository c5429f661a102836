"""The C++ host shim (include/bf_node_shim.hpp) driven like a reference node: yaml config in, one
jack_callback per period, /theta messages in between -- compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from beamform_amd.params import AIRA16_XY, make_params
from beamform_amd.synth import make_scene
from conftest import ROOT, rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo,interf", [("das", ()), ("mvdr", ()), ("lcmv", (-60.0, 90.0)), ("phasempf", ())])
def test_file_node_matches_oracle(tmp_path, algo, interf):
    import oracle
    exe = os.path.join(ROOT, "examples", "file_node")
    assert os.path.exists(exe), "run `make` first"
    M, F = 4, 24
    lines = ["verbose: true", "initial_angle: 15.0"]
    lines += [f"mic{i}: {{id: {i}, x: {x:.3f}, y: {y:.3f}, z: 0.000}}" for i, (x, y) in enumerate(AIRA16_XY[:M])]
    lines += [f"angle_interf{k + 1}: {a}" for k, a in enumerate(interf)] + [f"angle_interf{len(interf) + 1}: 181.0"]
    cfg = tmp_path / "beamform_config.yaml"
    cfg.write_text("\n".join(lines) + "\n")
    x = make_scene(M, F, seed=55)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    (tmp_path / "theta.txt").write_text("9 -35.0\n")
    subprocess.check_call([exe, algo, str(cfg), str(tmp_path / "in.f32"), str(tmp_path / "out.f32"), str(tmp_path / "theta.txt")])
    y = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    p = make_params(algo, n_mics=M, theta=15.0, interf=interf)
    node = oracle.OracleNode(p)
    y1, _ = node.process(np.ascontiguousarray(x[:, : 9 * 512]))
    node.set_theta(-35.0)
    y2, _ = node.process(np.ascontiguousarray(x[:, 9 * 512:]))
    y_ref = np.concatenate([y1, y2])
    ok = np.isfinite(y_ref)
    assert y.shape == y_ref.shape and (np.isfinite(y) == ok).all()
    assert rel_l2(y[ok], y_ref[ok]) < 1e-5
    if algo == "das":  # the default configuration computes in double like das.cpp:16-24 (bf_config_init: BF_DAS_F64): the float output is the oracle's
        assert np.array_equal(y, y_ref)


def test_theta_scan_picks_the_target_and_matches_oracle_energies(tmp_path):
    """examples/theta_scan: the controllers' "publish theta, listen 50 windows, take the RMS" (energy2theta.py:23-27,
    62-101) as one n_dirs batch per block.  Every candidate's RMS must equal the RMS of a reference node that has been
    steered to that angle since the start of the stream; the 20-degree target must win."""
    import oracle
    exe = os.path.join(ROOT, "examples", "theta_scan")
    assert os.path.exists(exe), "run `make` first"
    M, W, blocks, D = 8, 10, 3, 18  # candidates every 20 degrees: -180, -160, ..., 160
    lines = ["initial_angle: 0.0"] + [f"mic{i}: {{id: {i}, x: {x:.3f}, y: {y:.3f}, z: 0.000}}"
                                      for i, (x, y) in enumerate(AIRA16_XY[:M])]
    cfg = tmp_path / "beamform_config.yaml"
    cfg.write_text("\n".join(lines) + "\n")
    x = make_scene(M, W * blocks, seed=91, silent_frac=0.0)
    (tmp_path / "in.f32").write_bytes(x.tobytes())
    out = subprocess.run([exe, "das", str(cfg), str(tmp_path / "in.f32"), str(D), str(W)], capture_output=True, text=True,
                         check=True).stdout.strip().splitlines()
    assert len(out) == blocks
    thetas = [-180.0 + 360.0 * d / D for d in range(D)]
    p = make_params("das", n_mics=M)
    nodes = [oracle.OracleNode(dict(p, theta=t)) for t in thetas]
    for b, line in enumerate(out):
        head, tail = line.split("|")
        blk, best_theta, best_rms = head.split()
        rms = np.array([float(v) for v in tail.split()])
        seg = np.ascontiguousarray(x[:, b * W * 512:(b + 1) * W * 512])
        ref = np.array([np.sqrt(np.mean(nd.process(seg)[0].astype(np.float64) ** 2)) for nd in nodes])
        assert int(blk) == b and np.abs(rms - ref).max() < 1e-5 * ref.max()
        assert float(best_theta) == 20.0 and abs(float(best_rms) - rms.max()) < 1e-12
