"""Edge cases of the C ABI on the GPU: empty batches, bad arguments, maximum sizes, silent input."""
import ctypes as C

import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import rel_l2

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("das_impls")]
TOL = 1e-5  # north_star tolerance


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("algo", ["das", "mvdr", "phasempf", "gsc"])
def test_empty_batch_is_a_no_op(algo):
    """n_frames = 0 returns BF_OK and leaves the carried state alone (a JACK client that is not yet READY)."""
    import oracle
    from conftest import Beamformer
    torch = _torch()
    M, F = 4, 10
    p = make_params(algo, n_mics=M, theta=15.0)
    x = make_scene(M, F, seed=9)
    bf = Beamformer(p)
    y1 = bf.process(np.ascontiguousarray(x[:, : 4 * 512]))
    dummy = torch.zeros(16, device="cuda")
    bf.process_device(dummy.data_ptr(), 0, dummy.data_ptr())   # device entry point, zero frames
    assert bf._L.bf_process_batch(bf._h, x.ctypes.data, 0, x.ctypes.data) == 0  # host entry point, zero frames
    y2 = bf.process(np.ascontiguousarray(x[:, 4 * 512:]))
    y = np.concatenate([y1, y2])
    y_ref = oracle.OracleNode(p).process(x)[0]
    ok = np.isfinite(y_ref)
    assert (np.isfinite(y) == ok).all()
    assert rel_l2(y[ok], y_ref[ok]) < TOL


def test_bad_arguments_are_rejected_not_fatal():
    from beamform_amd.capi import BfError
    from conftest import Beamformer
    torch = _torch()
    p = make_params("das", n_mics=4)
    bf = Beamformer(p)
    L, h = bf._L, bf._h
    buf = torch.zeros(4 * 512, device="cuda")
    assert L.bf_process_batch_device(h, None, 1, buf.data_ptr(), None, None) == -22
    assert L.bf_process_batch_device(h, buf.data_ptr(), 1, None, None, None) == -22
    assert L.bf_process_batch_device(None, buf.data_ptr(), 1, buf.data_ptr(), None, None) == -22
    x = np.zeros((4, 256), np.float32)
    ptrs = (C.c_void_p * 4)(*[x[m].ctypes.data for m in range(4)])
    out = np.zeros(512, np.float32)
    assert L.bf_process_hop(h, ptrs, out.ctypes.data, 256) == -22      # nframes != configured hop
    assert b"hop" in L.bf_last_error(h)
    assert L.bf_set_state(h, b"\0" * 64, 64) == -22                     # too short / wrong header
    blob = bytearray(bf.get_state())
    blob[0] ^= 0xFF
    assert L.bf_set_state(h, bytes(blob), len(blob)) == -22             # bad magic
    with pytest.raises(BfError):
        bf.set_interference(1, 30.0)                                    # das has no interferers
    with pytest.raises(BfError):
        bf.set_theta_dir(1, 0.0)                                        # single-direction handle
    with pytest.raises(BfError):
        Beamformer(make_params("das", n_mics=4, hop=300))               # JACK periods are powers of two, 64 ... 4096
    # the handle still works afterwards
    y = bf.process(make_scene(4, 3, seed=1))
    assert np.isfinite(y).all()


@pytest.mark.parametrize("M", [9, 31, 32])
def test_das_maximum_microphone_counts(M):
    """BF_MAX_MICS = 32; > 8 microphones take the kernel variant whose gain tables stay in L2; odd counts pad the pair."""
    import oracle
    from conftest import Beamformer
    _torch()
    rng = np.random.default_rng(M)
    mics = [(0.0, 0.0)] + [tuple(rng.uniform(-0.25, 0.25, 2)) for _ in range(M - 1)]
    p = make_params("das", n_mics=M, mics=mics, theta=-25.0)
    x = make_scene(M, 20, seed=M, mics=p["mics"])
    y_ref = oracle.OracleNode(p).process(x)[0]
    y = Beamformer(p).process(x)
    assert rel_l2(y, y_ref) < TOL


def test_phase_with_32_microphones():
    import oracle
    from conftest import Beamformer
    _torch()
    rng = np.random.default_rng(5)
    M = 32
    mics = [(0.0, 0.0)] + [tuple(rng.uniform(-0.25, 0.25, 2)) for _ in range(M - 1)]
    p = make_params("phase", n_mics=M, mics=mics, theta=20.0)
    x = make_scene(M, 10, seed=6, mics=p["mics"])
    y_ref = oracle.OracleNode(p).process(x)[0]
    assert rel_l2(Beamformer(p).process(x), y_ref) < TOL


@pytest.mark.parametrize("algo", ["das", "mvdr", "lcmv", "gss", "phase", "phasempf", "mcra", "gsc"])
def test_digital_silence(algo):
    """All-zero input: every node must do what the reference does with it.  The magnitude gates of mvdr/lcmv/gss
    (mvdr.cpp:85) stay closed, so no zero covariance is inverted and every node emits exact zeros."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 4, 14
    p = make_params(algo, n_mics=M, interf=(-60.0,) if algo in ("lcmv", "gss") else ())
    x = np.zeros((M, F * 512), np.float32)
    y_ref = oracle.OracleNode(p).process(x)[0]
    y = Beamformer(p).process(x)
    assert (np.isfinite(y) == np.isfinite(y_ref)).all()
    ok = np.isfinite(y_ref)
    assert np.array_equal(y[ok], y_ref[ok])


def test_full_scale_square_wave_input():
    """+-1.0 full-scale, spectrally dense input (every odd harmonic): no overflow surprises in the fp32 path."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F = 8, 12
    t = np.arange(F * 512)
    x = np.stack([np.where(((t + 3 * m) // 37) % 2 == 0, 1.0, -1.0) for m in range(M)]).astype(np.float32)
    p = make_params("das", n_mics=M, theta=40.0)
    assert rel_l2(Beamformer(p).process(x), oracle.OracleNode(p).process(x)[0]) < TOL


def test_maximum_look_directions():
    """BF_MAX_DIRS = 64 beams from one input; spot-check a few against their own oracle node."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, F, D = 8, 24, 64
    thetas = list(np.linspace(-180.0, 175.0, D))
    p = make_params("das", n_mics=M)
    x = make_scene(M, F, seed=64)
    bf = Beamformer(p, n_dirs=D)
    bf.set_thetas(thetas)
    y = bf.process(x)
    assert y.shape == (D, F * 512)
    for d in (0, 17, 40, 63):
        assert rel_l2(y[d], oracle.OracleNode(dict(p, theta=thetas[d])).process(x)[0]) < TOL


def test_mvdr_many_streams_times_tiles_exceeds_65535_blocks():
    """tiles x streams goes into grid.x (2^31-1), not grid.y (65535): 40 streams x 2048-frame batches = 2560 tile-streams
    here; the launch geometry is the same code path as 64 streams x 65536 frames = 131072."""
    import oracle
    from conftest import Beamformer
    _torch()
    M, S, F = 4, 40, 96
    p = make_params("mvdr", n_mics=M, theta=10.0)
    xs = np.stack([make_scene(M, F, seed=300 + (s % 3)) * (0.5 + 0.5 * s / S) for s in range(S)]).astype(np.float32)
    y = Beamformer(p, n_streams=S).process(xs)
    for s in (0, 17, 39):
        y_ref = oracle.OracleNode(p).process(xs[s])[0]
        ok = np.isfinite(y_ref)
        assert (np.isfinite(y[s]) == ok).all()
        assert rel_l2(y[s][ok], y_ref[ok]) < TOL


@pytest.mark.parametrize("algo,layout", [("das", "planar"), ("das", "interleaved"), ("mvdr", "planar")])
def test_long_host_batch_is_pipelined_in_chunks(algo, layout):
    """bf_process_batch splits long single-stream batches into 8 chunks (copy / compute / copy overlap on three streams):
    the result must be the one-shot device-path result, with pageable and with page-locked host buffers."""
    from beamform_amd.capi import BF_INTERLEAVED, BF_PLANAR, host_array
    from conftest import Beamformer
    torch = _torch()
    M, F = 4, 8192 + 37  # ragged last chunk
    p = make_params(algo, n_mics=M, theta=25.0)
    x = make_scene(M, F, seed=4242)
    lay = BF_PLANAR if layout == "planar" else BF_INTERLEAVED
    xin = x if layout == "planar" else np.ascontiguousarray(x.T)
    ref_bf = Beamformer(p, layout=lay)
    xd = torch.from_numpy(xin).cuda()
    yd = torch.empty(F * 512, dtype=torch.float32, device="cuda")
    ref_bf.process_device(xd.data_ptr(), F, yd.data_ptr())
    torch.cuda.synchronize()
    y_dev = yd.cpu().numpy()
    y_pageable = Beamformer(p, layout=lay).process(xin)
    xp = host_array(xin.shape)
    xp[...] = xin
    yp = host_array((F * 512,))
    y_pinned = Beamformer(p, layout=lay).process(xp, out=yp)
    for y in (y_pageable, y_pinned):
        if algo == "das":  # the fused kernel is bit-independent of how a stream is cut into batches and runs
            assert np.array_equal(y, y_dev)
        else:              # mvdr rebuilds the covariance from history at every tile start: last-bit differences in double
            ok = np.isfinite(y_dev)
            assert (np.isfinite(y) == ok).all() and rel_l2(y[ok], y_dev[ok]) < 1e-6
