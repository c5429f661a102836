"""Several look directions from one set of forward transforms (das_fused_dirs_kernel: planar input, <= 8 microphones, >= 6
directions, no spectrum dump).  Every direction must equal the single-direction node steered to that angle (das_fused_kernel) to
the last bits -- same transforms, same order of accumulation; hipcc fuses the analysis window's products into the first
butterflies differently in the two kernels, which moves single results by one unit in the last place -- and the oracle within the
north_star tolerance.  Batch cuts inside the kernel itself change nothing, bit for bit."""
import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("das_f32")]
TOL_TIME = 1e-5


def _same(y, ref, x):
    """equal up to the last bits of float32 arithmetic: no sample differs by more than 4e-7 of the INPUT's peak (a transform's
    rounding error scales with the frame, not with the sample: the fade-in of a first hop holds samples of 1e-6 beside an error
    of 1e-8), and over more than a few hops the relative L2 stays below 1e-6"""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    ok = float(np.abs(y - ref).max()) <= 4e-7 * float(np.abs(x).max())
    if y.size >= 4 * 512:
        ok = ok and rel_l2(y, ref) < 1e-6
    return ok


def _run(p, x, thetas, cuts=None, n_streams=1):
    import torch
    from conftest import Beamformer
    assert torch.cuda.is_available()
    F = x.shape[-1] // 512
    D = len(thetas)
    bf = Beamformer(p, n_streams=n_streams, n_dirs=D)
    bf.set_thetas(thetas)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    cuts = cuts or [0, F]
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        xs = xd[..., a * 512:b * 512].contiguous()
        yd = torch.empty((n_streams * D, (b - a) * 512), dtype=torch.float32, device="cuda")
        bf.process_device(xs.data_ptr(), b - a, yd.data_ptr())
        torch.cuda.synchronize()
        outs.append(yd.cpu().numpy())
    return np.concatenate(outs, axis=1)


def _single(p, x, theta):
    from conftest import Beamformer
    return Beamformer(dict(p, theta=theta)).process(x)


@pytest.mark.parametrize("M,F,D", [(8, 200, 16), (8, 37, 7), (3, 50, 6), (5, 66, 9), (8, 48, 20), (4, 1, 8), (1, 20, 6), (2, 33, 16), (7, 19, 64)])
def test_shared_forward_transforms_equal_the_single_direction_node(M, F, D):
    p = make_params("das", n_mics=M, theta=0.0)
    x = make_scene(M, F, seed=1300 + M + D)
    thetas = [float(v) for v in np.linspace(-170.0, 175.0, D)]
    y = _run(p, x, thetas)
    import oracle
    for d in (0, 1, D // 2, D - 2, D - 1):
        ref = _single(p, x, thetas[d])
        assert _same(y[d], ref, x), (d, float(np.abs(y[d] - ref).max()))
    y_ref, _ = oracle.OracleNode(dict(p, theta=thetas[D - 1])).process(x)
    assert rel_l2(y[D - 1], y_ref) < TOL_TIME


def test_shared_forward_transforms_across_uneven_batches_and_streams():
    """Carried state (last input hop, one overlap-add tail per output stream) across batches of odd length -- a lone frame A in the
    last round, a one-frame batch -- and two input streams."""
    M, F, D, S = 8, 61, 8, 2
    p = make_params("das", n_mics=M, theta=0.0)
    x = np.stack([make_scene(M, F, seed=1400), make_scene(M, F, seed=1401)])
    thetas = [float(v) for v in np.linspace(-90.0, 90.0, D)]
    y = _run(p, x, thetas, cuts=[0, 7, 8, 9, 30, 61], n_streams=S)
    whole = _run(p, x, thetas, n_streams=S)
    assert np.array_equal(y, whole)
    for s in range(S):
        for d in (0, 3, 7):
            ref = _single(p, x[s], thetas[d])
            assert _same(y[s * D + d], ref, x)


def test_shared_forward_transforms_at_the_baseline_size():
    """65 536 frames, 16 directions: every run boundary (two atomic adds into a zeroed hop) against the single-direction node."""
    import torch
    from conftest import Beamformer
    M, F, D = 8, 65536, 16
    p = make_params("das", n_mics=M, theta=0.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    xd = torch.rand((1, M, F * 512), device="cuda", generator=g) - 0.5
    thetas = [float(v) for v in np.linspace(-180.0, 157.5, D)]
    bf = Beamformer(p, n_dirs=D)
    bf.set_thetas(thetas)
    yd = torch.empty((D, F * 512), dtype=torch.float32, device="cuda")
    bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    torch.cuda.synchronize()
    for d in (0, 5, 15):
        b1 = Beamformer(dict(p, theta=thetas[d]))
        y1 = torch.empty((1, F * 512), dtype=torch.float32, device="cuda")
        b1.process_device(xd.data_ptr(), F, y1.data_ptr())
        torch.cuda.synchronize()
        assert _same(yd[d].cpu().numpy(), y1[0].cpu().numpy(), np.float32(0.5))


def test_block_per_direction_path_still_serves_many_directions():
    """BF_DAS_SHARED_DIRS=0 (read once per process, hence a child): eight directions through das_fused_kernel, one block per
    direction, against the oracle."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import torch, oracle
from conftest import Beamformer_f32 as Beamformer   # look directions that share their forward transforms: the fp32 opt-in
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
M, F, D = 8, 40, 8
p = make_params("das", n_mics=M, theta=0.0)
x = make_scene(M, F, seed=1500)
thetas = [float(v) for v in np.linspace(-160.0, 160.0, D)]
bf = Beamformer(p, n_dirs=D); bf.set_thetas(thetas)
xd = torch.from_numpy(x).cuda()
yd = torch.empty((D, F * 512), dtype=torch.float32, device="cuda")
bf.process_device(xd.data_ptr(), F, yd.data_ptr()); torch.cuda.synchronize()
y = yd.cpu().numpy()
worst = 0.0
for d in (0, 3, 7):
    y_ref, _ = oracle.OracleNode(dict(p, theta=thetas[d])).process(x)
    worst = max(worst, rel_l2(y[d], y_ref))
print("RESULT " + json.dumps({"worst": worst}))
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BF_DAS_SHARED_DIRS="0"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["worst"] < TOL_TIME, res
