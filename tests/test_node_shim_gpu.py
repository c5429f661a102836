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
