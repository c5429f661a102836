"""CPU suite (no GPU): host-side logic of the product -- the kernels' algebra run through the CPU
emulation harness, the C ABI surface, the YAML reader, and the loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
from conftest import ROOT, rel_l2

P = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(P)


def test_emulated_fft32_and_fft1024(emul_lib):
    rng = np.random.default_rng(0)
    for dit in (0, 1):
        for d in (-1, 1):
            x = rng.standard_normal(32) + 1j * rng.standard_normal(32)
            o = np.empty(32, np.complex128)
            emul_lib.emul_fft32(_ptr(x), _ptr(o), d, dit)
            ref = np.fft.fft(x) if d < 0 else np.fft.ifft(x) * 32
            assert rel_l2(o, ref) < 1e-14
    for use_float, tol in ((0, 1e-14), (1, 5e-7)):
        for d in (-1, 1):
            x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
            o = np.empty(1024, np.complex128)
            emul_lib.emul_fft1024(_ptr(x), _ptr(o), d, use_float)
            ref = np.fft.fft(x) if d < 0 else np.fft.ifft(x) * 1024
            assert rel_l2(o, ref) < tol


def test_emulated_small_and_split_transforms(emul_lib):
    """fft_small.hpp (N = 32 x NL: 1024 / N frames side by side through one transpose plane, an NL-point second pass per frame -- the transforms of
    stft_small_kernel / istft_small_kernel) and the radix-2 steps around FFT-1024 that istft_split_kernel / stft_bins_split_kernel use at N = 2048,
    instantiated on the CPU against numpy: index maps, twiddles, the one-transform backward path of a real frame."""
    rng = np.random.default_rng(7)
    for n in (512, 256, 128):
        g = 1024 // n
        for d in (-1, 1):
            x = rng.standard_normal((g, n)) + 1j * rng.standard_normal((g, n))
            o = np.empty((g, n), np.complex128)
            assert emul_lib.emul_fft_small(n, _ptr(x), _ptr(o), d) == 0
            ref = np.fft.fft(x, axis=1) if d < 0 else np.fft.ifft(x, axis=1) * n
            assert rel_l2(o, ref) < 1e-14
    x = rng.standard_normal(2048) + 1j * rng.standard_normal(2048)
    o = np.empty(2048, np.complex128)
    emul_lib.emul_fft2048_split(_ptr(x), _ptr(o), -1)
    assert rel_l2(o, np.fft.fft(x)) < 1e-14
    yr = rng.standard_normal(2048)           # backward: the spectrum of a real frame comes back from ONE complex FFT-1024
    Y = np.fft.fft(yr)
    back = np.empty(2048)
    emul_lib.emul_fft2048_split(_ptr(Y), _ptr(back), 1)
    assert rel_l2(back, yr * 2048) < 1e-14


def test_emulated_fft2048_on_a_full_wavefront(emul_lib):
    """N = 2048 = 32 registers x 64 lanes (das_fused_wave2048_kernel): the two-half transpose plane read as one 64-column transform and the
    radix-2 stage between the halves of the wavefront, both directions, against numpy."""
    rng = np.random.default_rng(11)
    for use_float, tol in ((0, 1e-14), (1, 5e-7)):
        for d in (-1, 1):
            x = rng.standard_normal(2048) + 1j * rng.standard_normal(2048)
            o = np.empty(2048, np.complex128)
            emul_lib.emul_fft2048_wave(_ptr(x), _ptr(o), d, use_float)
            ref = np.fft.fft(x) if d < 0 else np.fft.ifft(x) * 2048
            assert rel_l2(o, ref) < tol


def test_emulated_fft1024_w64(emul_lib):
    """64-lane x 16-point three-pass factorisation (fft1024_w64.hpp): index maps and twiddles."""
    rng = np.random.default_rng(1)
    for use_float, tol in ((0, 1e-14), (1, 5e-7)):
        for d in (-1, 1):
            x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
            o = np.empty(1024, np.complex128)
            emul_lib.emul_fft1024_w64(_ptr(x), _ptr(o), d, use_float)
            ref = np.fft.fft(x) if d < 0 else np.fft.ifft(x) * 1024
            assert rel_l2(o, ref) < tol
    for d in (-1, 1):  # the fp64 kernel's exchange: segments rotated by 4 b columns, the shift absorbed by the tw2' table
        x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
        o = np.empty(1024, np.complex128)
        emul_lib.emul_fft1024_w64_rot(_ptr(x), _ptr(o), d)
        ref = np.fft.fft(x) if d < 0 else np.fft.ifft(x) * 1024
        assert rel_l2(o, ref) < 1e-14


@pytest.mark.parametrize("M,theta", [(8, 20.0), (4, 0.0), (3, -75.0), (16, 135.0), (1, 0.0)])
def test_emulated_fused_das_matches_oracle(emul_lib, M, theta):
    """Pair packing + Hermitian-part gains + unpaired inverse, in fp32 exactly as the kernel does it."""
    import oracle
    F = 10
    p = make_params("das", n_mics=M, theta=theta)
    x = make_scene(M, F, seed=31 + M)
    y_ref, _ = oracle.OracleNode(p).process(x)
    y = np.empty(F * 512, np.float32)
    mx = np.array([m[0] for m in p["mics"]])
    my = np.array([m[1] for m in p["mics"]])
    emul_lib.emul_das_fused(M, 512, C.c_double(48000.0), _ptr(mx), _ptr(my), C.c_double(theta), _ptr(x), C.c_long(F), _ptr(y))
    assert rel_l2(y, y_ref) < 1e-6


@pytest.mark.parametrize("M,theta,F", [(8, 20.0, 9), (5, -110.0, 6), (1, 0.0, 3)])
def test_emulated_frame_pair_das_matches_oracle(emul_lib, M, theta, F):
    """das_f64_pair_kernel's formulation on the CPU with the kernel's own gain table and addressing (geometry.hpp das_mic_gains_w64_f64:
    bins 0 .. 512 in rows of 65, mirror bins read from row (3 - g, 3 - k3) column 64 - lane and conjugated): two frames of a microphone per
    transform, real / imaginary part of ONE backward transform = the two frames.  Odd frame counts end on a lone frame."""
    import oracle
    p = make_params("das", n_mics=M, theta=theta)
    x = make_scene(M, F, seed=77 + M)
    y_ref, _ = oracle.OracleNode(p).process(x)
    y = np.empty(F * 512, np.float32)
    mx = np.array([m[0] for m in p["mics"]])
    my = np.array([m[1] for m in p["mics"]])
    emul_lib.emul_das_pair_f64(M, C.c_double(48000.0), _ptr(mx), _ptr(my), C.c_double(theta), _ptr(x), C.c_long(F), _ptr(y))
    assert rel_l2(y, y_ref) < 2e-7   # double arithmetic up to the float stores


@pytest.mark.parametrize("hop,M,F", [(256, 8, 11), (128, 3, 14), (64, 6, 21)])
def test_emulated_frame_interleaving_das_matches_oracle(emul_lib, hop, M, F):
    """das_fused_small_kernel's formulation on the CPU: 1024 / N consecutive frames interleaved into one 1024-point sequence, the N-point
    pair gains repeated (geometry.hpp das_pair_gains_interleaved) -- the N-point chain of every frame is the 1024-point chain of the
    interleaved sequence.  Frame counts that are not multiples of the group size end on a partial group."""
    import oracle
    p = make_params("das", n_mics=M, theta=40.0, hop=hop)
    x = make_scene(M, F, hop=hop, seed=5 + hop)
    y_ref, _ = oracle.OracleNode(p).process(x)
    y = np.empty(F * hop, np.float32)
    mx = np.array([m[0] for m in p["mics"]])
    my = np.array([m[1] for m in p["mics"]])
    emul_lib.emul_das_small(M, hop, C.c_double(48000.0), _ptr(mx), _ptr(my), C.c_double(40.0), _ptr(x), C.c_long(F), _ptr(y))
    assert rel_l2(y, y_ref) < 1e-6   # fp32 gain table


def test_host_geometry_matches_oracle(emul_lib):
    import oracle
    p = make_params("das", n_mics=8, theta=57.0)
    node = oracle.OracleNode(p)
    f = np.empty(1024)
    emul_lib.emul_freqs(1024, C.c_double(48000.0), _ptr(f))
    assert np.array_equal(f, node.freqs())
    h = np.empty(1024)
    emul_lib.emul_hann(1024, _ptr(h))
    assert np.array_equal(h, node.hann())
    mx = np.array([m[0] for m in p["mics"]])
    my = np.array([m[1] for m in p["mics"]])
    tau = np.empty(8)
    emul_lib.emul_delays(8, _ptr(mx), _ptr(my), C.c_double(57.0), _ptr(tau))
    assert np.array_equal(tau, node.delays())


def test_library_exports_every_declared_symbol():
    from beamform_amd import capi
    lib = capi.load()
    header = open(os.path.join(ROOT, "include", "bfcore.h")).read()
    declared = set(re.findall(r"\b(bf_[a-z_0-9]+)\s*\(", header))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"bfcore" in lib.bf_version()


def test_config_defaults_are_the_launch_file_values():
    from beamform_amd import capi
    lib = capi.load()
    c = capi.BfConfig()
    assert lib.bf_config_init(C.byref(c), 1) == 0       # mvdr.launch:6-10
    assert (c.past_windows, c.freq_mag_threshold, c.freq_max, c.freq_min, c.out_amp) == (10, 0.001, 16000.0, 100.0, 1.0)
    assert (c.n_mics, c.hop, c.sample_rate) == (3, 512, 48000.0)   # aira3, beamform_config.yaml:15-17
    assert (c.mic_x[2], c.mic_y[2]) == (-0.156, -0.090)
    assert lib.bf_config_init(C.byref(c), 5) == 0       # phasempf.launch
    assert (c.min_phase, c.min_mag, c.smooth_size, c.mcra_L, c.out_amp) == (30.0, 0.05, 3, 50, 2.5)
    assert lib.bf_config_init(C.byref(c), 3) == 0       # gss.launch
    assert (c.out_amp, c.mu, c.lambda_) == (0.1, 0.001, 0.0)
    assert lib.bf_config_init(C.byref(c), 4) == 0       # phase: fallbacks of phase.cpp:180,187 (Q14)
    assert (c.min_phase, c.mag_mult, c.mag_threshold) == (10.0, 0.1, 0.05)
    assert lib.bf_config_init(C.byref(c), 9) != 0


YAML = b"""verbose: true
initial_angle: 12.5

#aira3
#mic0: {id: 0, x:  9.000, y:  9.000}
mic0:  {id:  0, x:  0.158, y:  0.115, z:  0.000}
mic1:  {id:  1, x:  0.158, y: -0.115, z:  0.000}
mic2:  {id:  2, x: -0.045, y:  0.000, z:  0.000}
mic3:  {id:  3, x: -0.050, y: -0.188, z:  0.000}
mic5:  {id:  5, x: 1, y: 1}

angle_interf1:  -60.0
angle_interf2:  90
angle_interf3:  181.0
angle_interf4:  10.0
past_windows: 12
freq_max: 8000
MCRA_L: 20
lambda: 0.5
write_file_path: ''
"""


def test_yaml_reader_follows_handle_params():
    """util.h:82-113: mics consumed until the first missing index, interferers until the first |angle| > 180."""
    from beamform_amd import capi
    lib = capi.load()
    c = capi.BfConfig()
    lib.bf_config_init(C.byref(c), 2)
    assert lib.bf_config_parse_yaml(C.byref(c), YAML) == 0
    assert c.theta == 12.5 and c.verbose == 1
    assert c.n_mics == 4 and (c.mic_x[3], c.mic_y[3]) == (-0.050, -0.188)      # mic4 missing -> mic5 ignored
    assert c.n_interf == 2 and (c.interf_angle[0], c.interf_angle[1]) == (-60.0, 90.0)
    assert (c.past_windows, c.freq_max, c.mcra_L, c.lambda_) == (12, 8000.0, 20, 0.5)
    path = os.path.join(ROOT, "tests", "golden", "_tmp_cfg.yaml")
    with open(path, "wb") as f:
        f.write(YAML)
    try:
        c2 = capi.BfConfig()
        lib.bf_config_init(C.byref(c2), 0)
        assert lib.bf_config_load_yaml(C.byref(c2), path.encode()) == 0 and c2.n_mics == 4
        assert lib.bf_config_load_yaml(C.byref(c2), b"/nonexistent.yaml") == -2
    finally:
        os.remove(path)


def test_create_fails_loudly_without_gpu_or_with_bad_config():
    import torch
    from beamform_amd import capi
    lib = capi.load()
    c = capi.config_from_params(make_params("das", n_mics=8))
    h = C.c_void_p()
    if not torch.cuda.is_available():
        assert lib.bf_device_count() <= 0
        assert lib.bf_create(C.byref(c), C.byref(h)) == -19           # BF_ENODEV: no CPU fallback
        assert b"no CPU fallback" in lib.bf_last_error(None)
        with pytest.raises(capi.BfError):
            capi.Beamformer(make_params("das", n_mics=8))
    c.hop = 384
    assert lib.bf_create(C.byref(c), C.byref(h)) in (-38, -19)        # not a power-of-two JACK period (or no device)
    c.hop, c.n_mics = 512, 0
    assert lib.bf_create(C.byref(c), C.byref(h)) == -22
    assert lib.bf_create(None, C.byref(h)) == -22
    assert lib.bf_strerror(-19).startswith(b"no usable HIP device")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: no product source may reference it."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "beamform_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle|liboracle|bf_oracle|np_oracle", txt, re.M):
                    bad.append(fn)
    assert not bad, bad


def test_das_f64_chunk_plan_tiles_every_stream(emul_lib):
    """The work queue of das_f64_pair_kernel (csrc/das_f64_plan.hpp, the host arithmetic das_f64_sched_kernel runs per chunk): for any
    batch shape and plan the chunks must tile every stream exactly once, start on even frames (so that WHICH frames share a transform
    never depends on the plan), hold at most 1000 pairs (the 10-bit fields of the kernel's work word), fit the table, and come longest
    first inside a stream's level order."""
    import ctypes as C
    rng = np.random.default_rng(5)
    shapes = [(65536, 1, 256), (65537, 1, 256), (1, 1, 256), (2, 1, 256), (7, 3, 256), (256, 256, 256), (300, 5, 256), (1_000_000, 1, 256),
              (4_000_000, 1, 256), (20_000, 300, 256), (2049, 1, 256), (16400, 1, 256), (33, 2, 304), (100_000, 7, 64)]
    shapes += [(int(rng.integers(1, 300_000)), int(rng.integers(1, 40)), int(rng.choice([64, 256, 304]))) for _ in range(40)]
    cap = 16384
    st, t0, n = (C.c_int * cap)(), (C.c_long * cap)(), (C.c_long * cap)()
    grid = C.c_int()
    emul_lib.emul_das_plan.restype = C.c_int
    emul_lib.emul_das_plan.argtypes = [C.c_long, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for env in (b"", b"0", b"5,3,1", b"104,8,4,2", b"40,7,5,3", b"1"):
        for F, S, cus in shapes:
            k = emul_lib.emul_das_plan(F, S, cus, env, cap, st, t0, n, C.byref(grid))
            assert 0 < k <= cap, (F, S, cus, env, k)
            assert 1 <= grid.value <= min(k, cus)
            seen = {s: [] for s in range(S)}
            for i in range(k):
                assert 0 <= st[i] < S and n[i] >= 1 and t0[i] % 2 == 0 and (n[i] + 1) // 2 <= 1000, (F, S, env, i, st[i], t0[i], n[i])
                seen[st[i]].append((t0[i], n[i]))
            for s in range(S):
                pos = 0
                for a, b in sorted(seen[s]):
                    assert a == pos, (F, S, env, s, a, pos)      # no gap, no overlap
                    assert b % 2 == 0 or a + b == F               # only a stream's last chunk ends on a lone frame
                    pos = a + b
                assert pos == F, (F, S, env, s, pos)
            if env == b"" and k > grid.value:   # level-major order: the table starts with one long chunk per block-share of every stream
                first = [n[i] for i in range(min(grid.value, k))]
                assert min(first) >= max(n[i] for i in range(k - 1, k)), (F, S)


def test_bench_tables_in_the_docs_match_their_records():
    """BASELINE.md section 4 and README.md carry ONE generated table of measured figures each (tools/bench_tables.py): regenerated from
    the records the table itself names (the driver's BENCH_rNN.json, this build's profiles/rNN_*bench*.json) it must come out the same
    text, i.e. no figure in it was typed or edited by hand."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_tables.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
