"""stft_bins_w64_kernel (das fp64, phase, phasempf: STFT and per-bin stage in one launch, spectra in LDS) against the oracle
over the shapes its index arithmetic distinguishes: odd microphone counts (idle pair slots), 4-microphone blocks (four frames per
round), frame counts that do not fill a round, several streams, interleaved input, hop-by-hop streaming; and against the
two-kernel chain (BF_FUSED_BINS=0, a separate process: the switch is read once)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2
from test_pipeline_gpu import check, run_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _node(algo, M, theta, **over):
    p = make_params("das" if algo == "das_f64" else algo, n_mics=M, theta=theta)
    if algo == "phase":
        p["mag_threshold"] = 0.0005        # open the magnitude gate on part of the bins of the test scene
    p.update(over)
    return p


@pytest.mark.parametrize("algo", ["das_f64", "phase", "phasempf"])
@pytest.mark.parametrize("M,F", [(8, 33), (7, 5), (5, 18), (4, 27), (3, 9), (2, 40), (1, 6), (8, 1), (4, 3)])
def test_fused_matches_oracle(algo, M, F):
    import oracle
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32
    p = _node(algo, M, 25.0)
    x = make_scene(M, F, seed=900 + 10 * M + F)
    y_ref, Y_ref = oracle.OracleNode(p).process(x, want_spectrum=True)
    y, Y = run_gpu(p, x, das_impl=BF_DAS_F64 if algo == "das_f64" else BF_DAS_FUSED_F32)
    check(y, Y, y_ref, Y_ref)


@pytest.mark.parametrize("algo", ["das_f64", "phase", "phasempf"])
def test_fused_streams_and_interleaved(algo):
    import oracle
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32, BF_INTERLEAVED
    M, F, S = 6, 11, 3
    p = _node(algo, M, -40.0)
    impl = BF_DAS_F64 if algo == "das_f64" else BF_DAS_FUSED_F32
    xs = np.stack([make_scene(M, F, seed=950 + s) for s in range(S)])           # [S, M, T]
    refs = [oracle.OracleNode(p).process(xs[s], want_spectrum=True) for s in range(S)]
    y, Y = run_gpu(p, xs, n_streams=S, F=F, das_impl=impl)
    yi, Yi = run_gpu(p, np.ascontiguousarray(xs.transpose(0, 2, 1)), n_streams=S, F=F, das_impl=impl, layout=BF_INTERLEAVED)
    for s in range(S):
        check(y[s], Y[s], *refs[s])
        check(yi[s], Yi[s], *refs[s])
        assert np.array_equal(y[s], yi[s])          # same arithmetic whatever the input layout


@pytest.mark.parametrize("algo", ["phase", "phasempf"])
def test_fused_streaming_equals_batch(algo):
    from beamform_amd.capi import Beamformer
    M, F = 8, 14
    p = _node(algo, M, 10.0)
    x = make_scene(M, F, seed=977)
    whole = Beamformer(p).process(x)
    bf = Beamformer(p)
    parts = [bf.process(x[:, :512 * 5]), bf.process(x[:, 512 * 5:512 * 6]), bf.process(x[:, 512 * 6:])]   # 5 + 1 + 8 frames
    assert np.array_equal(np.concatenate(parts), whole)


CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch
from beamform_amd.capi import Beamformer, BF_DAS_F64, BF_DAS_FUSED_F32
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
out = {}
for algo, M, F in [("das", 8, 21), ("phase", 8, 21), ("phasempf", 5, 30), ("phase", 4, 13)]:
    p = make_params(algo, n_mics=M, theta=33.0)
    if algo == "phase":
        p["mag_threshold"] = 0.0005
    x = make_scene(M, F, seed=990 + M)
    out[f"{algo}{M}"] = Beamformer(p, das_impl=BF_DAS_F64 if algo == "das" else BF_DAS_FUSED_F32).process(x)
np.savez(sys.argv[1], **out)
"""


def test_fused_equals_the_two_kernel_chain(tmp_path):
    """BF_FUSED_BINS=2 (stft + per-bin stage in one kernel, spectra in LDS: stft_bins_w64_kernel; the default for phase / phasempf, for
    das only when the one-launch kernel is switched off) against =0 (two kernels, spectra in HBM).  The fused kernel runs the 64-lane
    transform, whose rounding differs from the chain's 32 x 32 transform at 1e-16: equal up to the last bit of the float output.
    BF_FUSED_BINS=1 differs from =2 for das only (the one-launch kernels, checked against the oracle below)."""
    res = {}
    for tag, env in (("0", dict(BF_FUSED_BINS="0")), ("w2", dict(BF_FUSED_BINS="2")), ("w1", dict(BF_FUSED_BINS="1"))):
        f = str(tmp_path / f"out{tag}.npz")
        subprocess.check_call([sys.executable, "-c", CHILD % ROOT, f], env=dict(os.environ, **env))
        res[tag] = np.load(f)
    for k in res["0"].files:
        assert same_floats(res["w2"][k], res["0"][k]), k
        if not k.startswith("das"):
            assert np.array_equal(res["w1"][k], res["w2"][k]), k
        else:
            d = np.abs(res["w1"][k].astype(np.float64) - res["0"][k]).max()
            assert d <= 2e-7 * np.abs(res["0"][k]).max(), (k, d)


def same_floats(a, b):
    """Equal up to the rounding of the float stores (a few ulp of the signal's scale): two evaluation orders of the same double sums."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a.shape == b.shape and np.abs(a - b).max() <= 4e-7 * max(np.abs(b).max(), 1e-30)


GEOMETRIES = {
    # name: (x, y) per microphone.  The reference drops z (util.h:82-92): microphones 1 and 7 of its aira16 array (beamform_config.yaml:21,27)
    # share x, y and therefore their weight row for every look direction -- the first case is make_params' default 8-microphone geometry
    "aira16-first-8 (1 = 7)": None,
    "all distinct": [(0.158, 0.115), (0.158, -0.115), (-0.045, 0.0), (-0.05, -0.188), (-0.195, 0.0), (-0.057, 0.186), (0.18, 0.0), (0.056, -0.171)],
    "2 = 3 and 5 = 6 (one pair is merged, the other is not)": [(0.0, 0.0), (0.1, 0.02), (-0.05, 0.12), (-0.05, 0.12), (0.07, -0.11), (0.13, 0.09), (0.13, 0.09), (-0.2, 0.0)],
    "1 = 2 = 3": [(0.0, 0.0), (0.1, 0.1), (0.1, 0.1), (0.1, 0.1), (-0.1, 0.05)],
    "three microphones, 1 = 2": [(0.0, 0.0), (0.0, -0.18), (0.0, -0.18)],
    "co-located (every row is 1)": [(0.0, 0.0)] * 6,
    "symmetric about the look direction (theta = 0: y -> -y)": [(0.0, 0.0), (0.1, 0.07), (0.1, -0.07), (-0.12, 0.03), (-0.12, -0.03), (0.2, 0.0)],
}


@pytest.mark.parametrize("geo", list(GEOMETRIES))
def test_das_f64_microphones_with_identical_weight_rows_share_a_transform(geo):
    """das_f64_pair_kernel puts the first two microphones (other than the reference microphone) whose weight rows are bitwise identical into
    one forward transform: conj(w) FFT(h (x_a + x_b)), the sum formed in double.  Geometries with one such pair, several, a triple, none,
    rows that become identical only for the steered angle, and a re-steer that separates them again -- float output against the oracle
    (bit for bit: the sums differ from the reference's at 1e-16) and a cut stream."""
    import oracle
    from beamform_amd.capi import BF_DAS_F64, Beamformer, launch_trace
    mics = GEOMETRIES[geo]
    M = 8 if mics is None else len(mics)
    F = 21
    theta = 0.0 if "symmetric" in geo else 35.0
    p = make_params("das", n_mics=M, theta=theta, **({} if mics is None else {"mics": mics}))
    x = make_scene(M, F, seed=3300 + M, mics=p["mics"])
    node = oracle.OracleNode(p)
    bf = Beamformer(p, das_impl=BF_DAS_F64)
    w = bf.weights()[:, :, 0]
    dup = any(np.array_equal(w[:, a], w[:, b]) for a in range(1, M) for b in range(a + 1, M))
    assert dup == ("distinct" not in geo), geo
    with launch_trace() as tr:
        y1 = bf.process(np.ascontiguousarray(x[:, :9 * 512]))
    assert any("das_f64_pair_kernel" in k for k in tr.kernels), tr.kernels
    r1 = node.process(np.ascontiguousarray(x[:, :9 * 512]))[0]
    # a new look direction between the batches: rows that coincided for the old angle may differ now (and the other way round)
    bf.set_theta(-70.0)
    node.set_theta(-70.0)
    y2 = bf.process(np.ascontiguousarray(x[:, 9 * 512:]))
    r2 = node.process(np.ascontiguousarray(x[:, 9 * 512:]))[0]
    y, r = np.concatenate([y1, y2]), np.concatenate([r1, r2])
    assert rel_l2(y, r) < 1e-6
    # bit for bit wherever there is signal; where the output is the transforms' own rounding residue (silence: 1e-16 of the scene's scale)
    # the two evaluation orders leave different residues
    scale = float(np.abs(r).max())
    loud = np.abs(r) > 1e-9 * scale
    assert loud.mean() > 0.5 and np.array_equal(y[loud], r[loud]), float(np.abs(y.astype(np.float64) - r).max())
    assert np.abs(y[~loud].astype(np.float64) - r[~loud]).max(initial=0.0) < 1e-12 * scale


@pytest.mark.parametrize("M,F,S", [(8, 33, 1), (7, 5, 1), (5, 18, 2), (4, 27, 1), (3, 9, 3), (2, 40, 1), (1, 6, 1), (8, 1, 1), (6, 700, 1)])
def test_das_f64_one_launch_matches_oracle(M, F, S):
    """das_f64_pair_kernel / das_f64_w64_kernel<1> (BF_DAS_F64 without a spectrum dump): the time output against the oracle, odd
    microphone counts, several streams, runs that recompute their first frame (F = 700 is cut into runs), and batch cuts."""
    import oracle
    from beamform_amd.capi import BF_DAS_F64, Beamformer
    p = make_params("das", n_mics=M, theta=-50.0)
    xs = np.stack([make_scene(M, F, seed=1200 + 7 * M + s) for s in range(S)])
    bf = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64)
    y = bf.process(xs if S > 1 else xs[0]).reshape(S, -1)
    for s in range(S):
        y_ref, _ = oracle.OracleNode(p).process(xs[s])
        assert rel_l2(y[s], y_ref) < 1e-6       # double arithmetic up to the float stores: far inside the 1e-5 budget
    # [sample][mic] input (the layout north_star names): 2, 4 or 8 microphones go through the same kernel body with the hops transposed
    # wavefront by wavefront into the blocks' rings (das_f64_ring_kernel); other counts are transposed on the device into a planar scratch
    # in front of das_f64_pair_kernel (interleaved_to_planar_kernel); one microphone: das_f64_w64_kernel<1> -- the same bytes out as for
    # planar input
    from beamform_amd.capi import BF_INTERLEAVED, launch_trace
    xi = np.ascontiguousarray(xs.transpose(0, 2, 1))
    bil = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
    with launch_trace() as tr:
        yi = bil.process(xi if S > 1 else xi[0]).reshape(S, -1)
    if M in (2, 4, 8):
        assert any("das_f64_ring_kernel" in k for k in tr.kernels) and not any("interleaved_to_planar_kernel" in k for k in tr.kernels), tr.kernels
        assert np.array_equal(yi, y)
    elif M >= 2:
        assert any("interleaved_to_planar_kernel" in k for k in tr.kernels) and any("das_f64_pair_kernel" in k for k in tr.kernels), tr.kernels
        assert np.array_equal(yi, y)
    else:
        assert same_floats(yi, y)
    if S == 1 and F >= 9:
        bi = Beamformer(p, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)   # carried hop in the interleaved layout across batch cuts
        cuts = [0, 2, F // 2, F]
        parts = [bi.process(np.ascontiguousarray(xi[0][a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
        assert same_floats(np.concatenate(parts), yi[0])   # which frames share a transform depends on the cuts (as for planar input)
        bi2 = Beamformer(p, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
        cuts = sorted({0, 2, 2 * (F // 4), F})
        parts = [bi2.process(np.ascontiguousarray(xi[0][a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), yi[0])   # even cuts keep every pair: bit for bit
    if S == 1 and F >= 9:
        bf2 = Beamformer(p, das_impl=BF_DAS_F64)
        cuts = [0, 1, 4, F // 2, F]
        parts = [bf2.process(np.ascontiguousarray(xs[0][:, a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
        assert same_floats(np.concatenate(parts), y[0])   # state (ring hop, tail) carries; which frames share a transform depends on the cuts
        # ... and cuts at EVEN frame counts leave every pair where it was: the hop across a cut is tail_in + head, the same float addition
        # as inside a batch -- bit for bit (the deterministic regression the planar kernel keeps)
        bf3 = Beamformer(p, das_impl=BF_DAS_F64)
        cuts = sorted({0, 2, 4, 2 * (F // 4), 2 * (F // 3), F})
        parts = [bf3.process(np.ascontiguousarray(xs[0][:, a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts), y[0])


def test_das_f64_one_launch_streaming_callbacks():
    """bf_process_hop on the fp64 das node: one callback at a time == batch."""
    from beamform_amd.capi import BF_DAS_F64, Beamformer
    M, F = 8, 12
    p = make_params("das", n_mics=M, theta=15.0)
    x = make_scene(M, F, seed=31)
    whole = Beamformer(p, das_impl=BF_DAS_F64).process(x)
    bf = Beamformer(p, das_impl=BF_DAS_F64)
    hops = [bf.process_hop(np.ascontiguousarray(x[:, t * 512:(t + 1) * 512])) for t in range(F)]
    assert same_floats(np.concatenate([np.asarray(h).reshape(-1) for h in hops]), whole)   # a callback is a lone frame, the batch pairs them


CHILD_FULL = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import torch
from beamform_amd.capi import Beamformer, BF_DAS_F64, BF_DAS_FUSED_F32
from beamform_amd.params import make_params
M, F = 8, 65536
g = torch.Generator(device="cuda").manual_seed(5)
x = (torch.rand((M, F * 512), device="cuda", generator=g) - 0.5) * 6.0     # loud enough to open phase's magnitude gate
y = torch.empty(F * 512, device="cuda")
for algo in ("das", "phase"):
    bf = Beamformer(make_params(algo, n_mics=M, theta=-15.0), das_impl=BF_DAS_F64 if algo == "das" else BF_DAS_FUSED_F32)
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print(algo, hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest(), float(y.abs().mean()))
    if len(sys.argv) > 1:
        np.save(sys.argv[1] + algo + ".npy", y.cpu().numpy())
"""


def test_fused_equals_two_kernel_chain_at_the_baseline_size(tmp_path):
    """BASELINE batch (8 microphones x 65 536 frames): the fused STFT + per-bin kernel (BF_FUSED_BINS=2: phase by default, das
    when the one-launch kernel is off; stft_bins_w64_kernel) against the two-kernel chain: equal up to the last bit of the float output."""
    outs = {}
    for tag, env in (("chain", dict(BF_FUSED_BINS="0")), ("fused64", dict(BF_FUSED_BINS="2"))):
        args = [sys.executable, "-c", CHILD_FULL % ROOT, str(tmp_path / tag)]
        r = subprocess.run(args, env=dict(os.environ, **env), capture_output=True, text=True, check=True)
        outs[tag] = [ln.split() for ln in r.stdout.strip().splitlines()]
    assert len(outs["chain"]) == 2
    assert all(float(ln[2]) > 1e-3 for ln in outs["chain"])
    for algo in ("das", "phase"):
        a, c = np.load(str(tmp_path / "fused64") + algo + ".npy"), np.load(str(tmp_path / "chain") + algo + ".npy")
        assert same_floats(a, c), algo


def test_das_f64_one_launch_at_the_baseline_size():
    """65 536 frames through das_f64_pair_kernel: oracle windows at random offsets and at the kernel's chunk boundaries (a frame's
    output hop depends on three input hops only), and equality with the fused fp32 kernel to float accuracy."""
    import oracle
    import torch
    from beamform_amd.capi import BF_DAS_F64, BF_DAS_FUSED_F32, Beamformer
    M, F, n = 8, 65536, 20
    p = make_params("das", n_mics=M, theta=35.0)
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
    y = torch.empty(F * 512, device="cuda")
    Beamformer(p, das_impl=BF_DAS_F64).process_device(x.data_ptr(), F, y.data_ptr())
    y32 = torch.empty(F * 512, device="cuda")
    Beamformer(p, das_impl=BF_DAS_FUSED_F32).process_device(x.data_ptr(), F, y32.data_ptr())
    torch.cuda.synchronize()
    assert ((y - y32).norm() / y.norm()).item() < 1e-6
    rng = np.random.default_rng(9)
    # chunk edges of the kernel's work queue at this size (das_f64_plan: 256 chunks of 104 pairs, then 8-, 4- and 2-pair chunks): windows
    # across the first edges of every level, the level switches and the end of the batch
    starts = [0, 3, 14, 15, 16, 17, 30, 31, 32, 33, 63, 64, 200, 207, 208, 400, 53230, 53248, 53260, 57340, 57344, 59390, 59392, 59400,
              F - 2 * n, F - n] + [int(v) for v in rng.integers(2, F - n, 8)]
    for t0 in starts:
        a = max(t0 - 2, 0)
        seg = x[:, a * 512:(t0 + n) * 512].cpu().numpy()
        y_ref, _ = oracle.OracleNode(p).process(np.ascontiguousarray(seg))
        got = y[t0 * 512:(t0 + n) * 512].cpu().numpy()
        assert rel_l2(got, y_ref[(t0 - a) * 512:]) < 1e-6, t0


def _sha1_of_das_f64_batch(reps, F=65536, M=8):
    import hashlib
    import torch
    from beamform_amd.capi import BF_DAS_F64, Beamformer
    p = make_params("das", n_mics=M, theta=35.0)
    g = torch.Generator(device="cuda").manual_seed(33)
    x = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
    y = torch.empty(F * 512, device="cuda")
    digests = []
    for rep in range(reps):
        bf = Beamformer(p, das_impl=BF_DAS_F64)   # a cold handle: the same carried state every time
        y.fill_(float("nan"))
        bf.process_device(x.data_ptr(), F, y.data_ptr())
        torch.cuda.synchronize()
        bf.close()
        digests.append(hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest())
    return digests


def test_das_f64_pair_kernel_is_deterministic():
    """The headline kernel hands frame pairs to wavefronts and chunks to blocks dynamically, completes the hop between two pairs
    first come first served and the hop between two chunks by two float atomic adds into a zeroed hop.  Which frames share a transform is
    fixed by the batch position and a + b == b + a, so every launch must produce the same BYTES: 32 launches of the 65 536-frame batch,
    sha1 of the output (a hop completed from a stale or missing partner half shows up as a different digest; NaN pre-fill catches a hop
    nobody stored)."""
    d = _sha1_of_das_f64_batch(32)
    assert len(set(d)) == 1, sorted(set(d))


def test_das_f64_pair_kernel_output_does_not_depend_on_the_chunk_plan():
    """Chunks start on even frames, so the pairs are the same whatever the plan; an edge inside a chunk is one float addition, an edge
    between chunks is 0 + a + b by atomics: the same float.  Static equal runs, the default guided plan and a plan of tiny chunks (every
    second hop completed by atomics, 8 wavefronts of a block spread over several chunks) must agree bit for bit."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_fused_bins_gpu as t; print(t._sha1_of_das_f64_batch(3, F=16400)[-1])"
            % (ROOT, os.path.join(ROOT, "tests")))
    out = {}
    for plan in ("0", "1", "3,2,1", "40,7,5,3"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BF_DAS_F64_SCHED=plan), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[plan] = r.stdout.strip().splitlines()[-1]
    assert len(set(out.values())) == 1, out


@pytest.mark.parametrize("M,F,S", [(8, 4099, 2), (4, 64, 3), (2, 1001, 2), (8, 700, 1), (8, 2, 1), (4, 1, 2)])
def test_das_f64_ring_kernel_is_the_planar_kernel_bit_for_bit(M, F, S):
    """das_f64_ring_kernel ([sample][mic] input, 2 / 4 / 8 microphones: the frame-pair kernel's body behind a transposition of every pair's
    two new hops into the block's ring of planar hop slots) against das_f64_pair_kernel on the same samples in the planar layout: the same
    bytes -- batches of several chunks per block (the ring wraps, generations of slot states), several streams (a chunk's first pair fills its
    wavefront's private slot), an odd frame count (a lone last frame has no second hop), the carried hop in the [sample][mic] layout
    across uneven batch cuts, and geometries with and without a merged microphone pair (the extra microphone's loads come from the ring too)."""
    from beamform_amd.capi import BF_DAS_F64, BF_INTERLEAVED, Beamformer, launch_trace
    for mics in (None, GEOMETRIES["all distinct"][:M] if M <= 8 else None):
        p = make_params("das", n_mics=M, theta=-50.0, **({"mics": mics} if mics else {}))
        xs = np.stack([make_scene(M, F, seed=3300 + 7 * M + s) for s in range(S)])
        xi = np.ascontiguousarray(xs.transpose(0, 2, 1))
        y = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64).process(xs if S > 1 else xs[0]).reshape(S, -1)
        bil = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
        with launch_trace() as tr:
            yi = bil.process(xi if S > 1 else xi[0]).reshape(S, -1)
        assert any("das_f64_ring_kernel" in k for k in tr.kernels), tr.kernels
        assert np.array_equal(yi, y)
        if S == 1 and F >= 9:   # batch cuts at even frame counts keep every pair where it was: bit for bit, the carried hop included
            cuts = sorted({0, 2, 2 * (F // 6), 2 * (F // 3), F})
            b2 = Beamformer(p, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
            parts = [b2.process(np.ascontiguousarray(xi[0][a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
            assert np.array_equal(np.concatenate(parts), yi[0])
