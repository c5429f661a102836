"""Every kernel an environment switch can select (DESIGN.md "Run-time switches") runs against the oracle: the switches are read
once per process, so each case is its own child process.  A variant that stays in libbfcore.so stays tested."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import torch, oracle
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
algo, M, interf, F, dump, precision = %(algo)r, %(M)d, %(interf)r, %(F)d, %(dump)r, %(precision)d
p = make_params(algo, n_mics=M, interf=interf, theta=20.0)
x = make_scene(M, F, seed=900 + M)
y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
bf = Beamformer(p, precision=precision)
xd = torch.from_numpy(x).cuda()
yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
Yd = torch.empty((F, 1024, 2), dtype=torch.float64, device="cuda")
bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr() if dump else 0)
torch.cuda.synchronize()
y = yd.cpu().numpy()
ok = np.isfinite(y_ref)
res = {"finite_mask_equal": bool((np.isfinite(y) == ok).all()), "time": rel_l2(y[ok], y_ref[ok])}
if dump:
    Y = Yd.cpu().numpy().view(np.complex128)[..., 0]
    if algo == "das":  # the fused kernel dumps the Hermitian part of y_fft (the part that reaches Re(ifft))
        Yh = 0.5 * (Y_ref + np.conj(np.roll(Y_ref[:, ::-1], 1, axis=1)))
        Y_ref = Yh
    fin = np.isfinite(Y_ref).all(axis=1)
    res["finite_frames_equal"] = bool((np.isfinite(Y).all(axis=1) == fin).all())
    res["spectrum"] = max(rel_l2(Y[t], Y_ref[t]) for t in range(F) if fin[t])
print("RESULT " + json.dumps(res))
"""

REF, MIXED = 0, 1   # bf_precision: complex doubles between the transforms (the default) / z48 spectra + fp32 backward transform
CASES = [
    # (environment, algo, mics, interferers, frames, spectrum dump, bf_config.precision)
    ({"BF_MVDR_GROUP": "1"}, "mvdr", 8, (), 30, True, MIXED),                 # group-per-problem kernel over LDS, z48 spectra
    ({"BF_MVDR_GROUP": "1"}, "lcmv", 8, (-60.0, 90.0), 30, False, MIXED),
    ({"BF_MVDR_GROUP": "1"}, "mvdr", 8, (), 30, True, REF),                   # ... reading complex doubles
    ({"BF_MVDR_GROUP": "1"}, "lcmv", 16, (-60.0, 90.0, 150.0), 24, True, REF),
    ({"BF_MVDR_TILE": "7"}, "lcmv", 8, (-60.0, 90.0), 40, True, MIXED),       # mvdr_fast_kernel<8, 3, false>: lanes straddle tiles, short last tile
    ({"BF_MVDR_TILE": "7"}, "mvdr", 8, (), 40, False, MIXED),                 # mvdr_fast_kernel<8, 1, false> + the fp32 backward transform
    ({"BF_MVDR_TILE": "7"}, "lcmv", 8, (-60.0, 90.0), 40, True, REF),         # ... <8, 3, true>
    ({"BF_MVDR_TILE": "7"}, "mvdr", 8, (), 40, False, REF),                   # ... <8, 1, true> + the fp64 backward transform (f64x2 rows)
    ({}, "mvdr", 3, (), 30, False, REF),                                      # <4, 1, true>: an odd microphone count's zero partner channel
    ({}, "mvdr", 3, (), 30, False, MIXED),
    ({}, "lcmv", 16, (-60.0, 90.0), 24, False, REF),                          # cov2d_kernel<4, 2, true> + the fp64 backward transform
    ({}, "lcmv", 16, (-60.0, 90.0, 150.0), 24, True, MIXED),                  # cov2d_kernel<4, 2, false>
    ({}, "mvdr", 12, (), 24, False, REF),                                     # cov2d_kernel<1, 3, true>
    ({}, "mvdr", 12, (), 24, False, MIXED),                                   # cov2d_kernel<1, 3, false> + istft32
    ({}, "phase", 8, (), 24, False, MIXED),                                   # fp32 backward transform wherever a per-bin stage can emit f32x2 rows
    ({}, "phasempf", 8, (), 24, False, MIXED),
    ({"BF_FUSED_BINS": "0"}, "das", 8, (), 24, False, MIXED),                 # das in double through the chain, f32x2 rows
    ({"BF_GSS_GROUP": "0"}, "gss", 8, (-60.0, 90.0), 30, True, REF),          # gss_lane_kernel<8, 4> (one lane per problem: the default from 57 streams on)
    ({"BF_GSS_GROUP": "0"}, "gss", 8, (), 24, True, REF),                     # ... <8, 1>: one source, the constraint term
    ({"BF_GSS_GROUP": "0"}, "gss", 3, (90.0,), 24, False, REF),               # ... <4, 4>, an odd microphone count
    ({"BF_GSS_GROUP": "1"}, "gss", 4, (-60.0, 90.0, 150.0), 24, False, REF),  # the group-per-problem kernel where the lane kernel would run
    ({"BF_GSC_SERIAL": "1"}, "gsc", 4, (), 10, False, REF),                   # gsc_nlms_kernel: the sums in the reference's tap order, one branch per lane
    ({"BF_GSC_SERIAL": "1"}, "gsc", 8, (), 8, False, REF),
    ({}, "gsc", 2, (), 8, False, REF),                                        # one blocking branch: gsc_nlms_par_kernel (one wavefront per stream)
    ({}, "gsc", 3, (), 8, False, REF),                                        # two branches: gsc_nlms_mw_kernel<2, ...>
    ({}, "gsc", 5, (), 8, False, REF),                                        # four branches: gsc_nlms_mw_kernel<4, ...>
]


@pytest.mark.parametrize("env,algo,M,interf,F,dump,precision", CASES,
                         ids=[f"{'_'.join(f'{k}={v}' for k, v in c[0].items()) or 'default'}-{c[1]}{c[2]}-k{len(c[3])}-{'mixed' if c[6] else 'ref'}{'-dump' if c[5] else ''}" for c in CASES])
def test_env_selected_kernel_matches_oracle(env, algo, M, interf, F, dump, precision):
    code = CHILD % dict(root=ROOT, algo=algo, M=M, interf=tuple(interf), F=F, dump=dump, precision=precision)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["finite_mask_equal"], res
    assert res["time"] < TOL, res
    if dump:
        assert res["finite_frames_equal"] and res["spectrum"] < TOL, res


CHILD_COND = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import torch, oracle
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
M, F = 8, 60
p = make_params("mvdr", n_mics=M, theta=20.0)
# one loud directional source, interferers and sensor noise 60 dB below it: the 10-frame covariance is numerically rank one and only the
# 1.001 diagonal loading (mvdr.cpp:239-243) keeps it invertible -- cond(R o whiteR) ~ 1e4 at every in-band bin
x = make_scene(M, F, seed=4242, sigma_s=0.3, sigma_i=3e-4, sigma_n=3e-4, silent_frac=0.0)
y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
bf = Beamformer(p, precision=int(sys.argv[2]))
xd = torch.from_numpy(x).cuda()
yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
Yd = torch.empty((F, 1024, 2), dtype=torch.float64, device="cuda")
bf.process_device(xd.data_ptr(), F, yd.data_ptr(), Yd.data_ptr())
torch.cuda.synchronize()
Y = Yd.cpu().numpy().view(np.complex128)[..., 0]
fin = np.isfinite(Y_ref).all(axis=1)
np.save(sys.argv[1], Y)
print("RESULT " + json.dumps({"spectrum": max(rel_l2(Y[t], Y_ref[t]) for t in range(F) if fin[t]), "frames": int(fin.sum())}))
"""


def test_z48_against_full_double_spectra_on_an_ill_conditioned_scene(tmp_path):
    """The packed 12-byte spectra of BF_PRECISION_MIXED (36 mantissa bits) against the default (full doubles in HBM) where it matters: a covariance that is
    rank one up to its diagonal loading.  Both stay inside the 1e-5 budget against the oracle; the figure that the packing itself
    costs is their mutual distance."""
    import numpy as np
    res, Ys = {}, {}
    for mode in ("1", "0"):
        f = str(tmp_path / f"Y{mode}.npy")
        out = subprocess.run([sys.executable, "-c", CHILD_COND % dict(root=ROOT), f, mode], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[mode] = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
        Ys[mode] = np.load(f)
    assert res["1"]["frames"] == res["0"]["frames"] > 40
    assert res["0"]["spectrum"] < TOL and res["1"]["spectrum"] < TOL, res
    fin = np.isfinite(Ys["0"]).all(axis=1)
    d = max(np.linalg.norm(Ys["1"][t] - Ys["0"][t]) / np.linalg.norm(Ys["0"][t]) for t in range(len(fin)) if fin[t] and np.abs(Ys["0"][t]).max() > 0)
    assert d < 1e-6, d          # 2^-37 on X times the condition number


CHILD_F64 = r"""
import sys, json, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import oracle
from beamform_amd.capi import BF_DAS_F64, BF_INTERLEAVED, Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
res = {}
for M, F in ((8, 700), (5, 37), (8, 1)):
    p = make_params("das", n_mics=M, theta=-25.0)
    x = make_scene(M, F, seed=640 + M)
    ref, _ = oracle.OracleNode(p).process(x)
    res[f"planar{M}x{F}"] = rel_l2(Beamformer(p, das_impl=BF_DAS_F64).process(x), ref)
    res[f"interleaved{M}x{F}"] = rel_l2(Beamformer(p, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED).process(np.ascontiguousarray(x.T)), ref)
# several streams through one launch: the chunk table is level-major over the streams (every stream's long chunk first), chunks never
# cross a stream, every stream starts from its own carried state; an odd frame count leaves every stream a lone last frame
S, M, F = 3, 6, 61
p = make_params("das", n_mics=M, theta=40.0)
xs = np.stack([make_scene(M, F, seed=700 + s) for s in range(S)])
ys = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64).process(xs).reshape(S, -1)
res["streams"] = max(rel_l2(ys[s], oracle.OracleNode(p).process(xs[s])[0]) for s in range(S))
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("env", [{}, {"BF_DAS_F64_SCHED": "0"}, {"BF_DAS_F64_SCHED": "5,3,1"}, {"BF_DAS_IL_RING": "0"}, {"BF_DAS_IL_RING": "1", "BF_DAS_F64_SCHED": "5,3,1"}],
                         ids=["default", "equal-chunks", "tiny-chunks", "transposer", "ring-tiny-chunks"])
def test_das_in_double_every_one_launch_kernel(env):
    """das at the reference's precision without a spectrum dump: das_f64_pair_kernel (planar input; its default chunk plan, equal static
    chunks and a plan of tiny chunks), [sample][mic] input through das_f64_ring_kernel (8 microphones: the hops transposed into the
    blocks' rings; tiny chunks: every second pair opens a chunk and fills its wavefront's private slot) and, BF_DAS_IL_RING=0 or another
    microphone count, through interleaved_to_planar_kernel in front of the planar kernel, against the oracle -- a batch that is cut into
    chunks, an odd microphone count with an odd number of frames, one lone frame."""
    out = subprocess.run([sys.executable, "-c", CHILD_F64 % dict(root=ROOT)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    assert len(res) == 7 and max(res.values()) < 1e-6, res


def test_bfcore_before_torch_shares_one_hip_runtime():
    """libbfcore.so loaded and used BEFORE torch is imported: capi.load() brings in torch's libamdhip64 first, so the later
    `import torch` finds the device (two HIP runtimes in one process: the second reports hipErrorNoDevice)."""
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
assert "torch" not in sys.modules
y = Beamformer(make_params("das", n_mics=4)).process(make_scene(4, 8, seed=1))
import torch
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(16, device="cuda", generator=g)
print("OK", y.shape, float(x.sum()) > 0)
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]
