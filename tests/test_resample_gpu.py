"""bf_resampler_* (csrc/resample.hip) against the oracle's restatement of libsamplerate's mono sinc converter."""
import numpy as np
import pytest

from beamform_amd import capi
from oracle.resample_oracle import SincResampler, resample_vectorised

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    """torch brings its own HIP runtime: it has to initialise before libbfcore's first HIP call in this process."""
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()


@pytest.mark.parametrize("rates", [(48000, 16000), (48000, 44100), (44100, 48000), (16000, 48000), (48000, 48000), (48000, 8000), (48000, 22050)])
def test_matches_oracle_period_by_period(rates):
    rng = np.random.default_rng(11)
    x = (0.2 * rng.standard_normal(9 * 512 + 131)).astype(np.float32)
    o = SincResampler(*rates)
    g = capi.Resampler(*rates)
    assert g.latency == o.half_len
    yo, yg = [], []
    for i in range(0, len(x), 512):                       # one call per JACK period, as convert_to_sample_rate
        yo.append(o.process(x[i:i + 512]))
        yg.append(g.process(x[i:i + 512]))
        assert len(yo[-1]) == len(yg[-1])
    yo, yg = np.concatenate(yo), np.concatenate(yg)
    assert len(yo) > 700
    # tolerance: 1e-6 of the signal scale (fp64 accumulation on both sides; the oracle's accumulated position drifts by ~1e-16/step)
    assert np.max(np.abs(yo - yg)) <= 1e-6 * np.max(np.abs(yo))


def test_cut_invariance_is_bit_exact():
    rng = np.random.default_rng(12)
    x = (0.3 * rng.standard_normal(20000)).astype(np.float32)
    whole = capi.Resampler(48000, 16000).process(x)
    for step in (1, 37, 512, 4096):
        g = capi.Resampler(48000, 16000)
        parts = [g.process(x[i:i + step]) for i in range(0, len(x), step)] if step > 1 else \
            [g.process(x[:5]), g.process(x[5:6]), g.process(x[6:6]), g.process(x[6:])]
        y = np.concatenate(parts)
        assert np.array_equal(y, whole)
    assert np.array_equal(whole, resample_vectorised(x, 48000, 16000)) or \
        np.max(np.abs(whole - resample_vectorised(x, 48000, 16000))) <= 1e-6 * np.max(np.abs(whole))


def test_reset_and_custom_table():
    rng = np.random.default_rng(13)
    x = (0.2 * rng.standard_normal(3000)).astype(np.float32)
    g = capi.Resampler(48000, 32000)
    a = g.process(x)
    g.reset()
    assert np.array_equal(g.process(x), a)
    # any half table of the same form: here a short triangular (linear-interpolation) kernel with 8 entries per zero crossing
    inc = 8
    tri = np.concatenate([1.0 - np.arange(inc) / inc, np.zeros(3)]).astype(np.float32)
    g.set_table(tri, inc)
    y = g.process(x)
    ref = resample_vectorised(x, 48000, 32000, coeffs=tri, index_inc=inc)
    assert len(y) == len(ref) and np.max(np.abs(y - ref)) <= 1e-6
    o = SincResampler(48000, 32000, coeffs=tri, index_inc=inc)
    assert np.max(np.abs(o.process(x) - y)) <= 1e-6


def test_errors():
    with pytest.raises(capi.BfError):
        capi.Resampler(48000, 100)                         # ratio below 1/256: src_is_valid_ratio fails (rosjack.cpp:170-172)
    with pytest.raises(capi.BfError):
        capi.Resampler(0, 16000)
    g = capi.Resampler(48000, 16000)
    import ctypes as C
    L = capi.load()
    x = np.zeros(4096, np.float32)
    out = np.zeros(8, np.float32)
    n = C.c_size_t()
    rc = L.bf_resampler_process(g._r, x.ctypes.data, x.size, out.ctypes.data, out.size, C.byref(n))
    assert rc == -22 and n.value == g.out_count(4096) > 8   # too little room: nothing consumed, the needed count reported
    assert len(g.process(x)) == n.value


def test_device_buffers_full_batch_properties():
    torch = pytest.importorskip("torch")
    F = 4096                                                # output hops of a 4096-frame batch
    n = np.arange(F * 512)
    x = (0.5 * np.sin(2 * np.pi * 997.0 * n / 48000) + 0.25 * np.sin(2 * np.pi * 3001.0 * n / 48000)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    g = capi.Resampler(48000, 16000)
    cap = g.out_count(x.size)
    yd = torch.empty(cap, device="cuda")
    got = g.process_device(xd.data_ptr(), x.size, yd.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert got == cap == (x.size - g.latency) * 16000 // 48000 + (1 if ((x.size - g.latency) * 16000) % 48000 else 0)
    y = yd.cpu().numpy()
    m = np.arange(cap)
    ref = 0.5 * np.sin(2 * np.pi * 997.0 * m / 16000) + 0.25 * np.sin(2 * np.pi * 3001.0 * m / 16000)
    err = (y - ref)[500:-500]
    assert np.sqrt(np.mean(err ** 2)) < 1e-4               # both tones are in the pass band: the output is the same signal at 16 kHz
    # linearity: resample(a + b) = resample(a) + resample(b) to fp32 rounding
    a = (0.5 * np.sin(2 * np.pi * 997.0 * n / 48000)).astype(np.float32)
    b = x - a
    ya = capi.Resampler(48000, 16000).process(a)
    yb = capi.Resampler(48000, 16000).process(b)
    assert np.max(np.abs((ya + yb) - y)) < 2e-6


def test_polyphase_and_generic_paths_agree():
    """48000 -> 16000 / 44100 run the tabulated (polyphase) kernel, 48000 -> 44101 has 44101 distinct positions and runs the
    generic one: both against the oracle."""
    rng = np.random.default_rng(21)
    x = (0.2 * rng.standard_normal(30000)).astype(np.float32)
    for rates in [(48000, 44101), (44100, 48000), (48000, 47999)]:
        y = capi.Resampler(*rates).process(x)
        ref = resample_vectorised(x, *rates)
        assert len(y) == len(ref) and np.max(np.abs(y - ref)) <= 1e-6 * np.max(np.abs(ref))
    # a finer table (3 x the entries per zero crossing) through both kernels
    from oracle.resample_oracle import default_table
    t = default_table()
    fine = np.interp(np.arange(3 * (len(t) - 1) + 1) / 3.0, np.arange(len(t)), t).astype(np.float32)
    for rates in [(48000, 44100), (48000, 44101)]:
        g = capi.Resampler(*rates)
        g.set_table(fine, 128 * 3)
        y = g.process(x)
        ref = resample_vectorised(x, *rates, coeffs=fine, index_inc=384)
        assert len(y) == len(ref) and np.max(np.abs(y - ref)) <= 1e-6 * np.max(np.abs(ref))


def test_handle_is_safe_against_a_second_thread():
    """process() from one thread while another polls latency / out_count and resets now and then: no crash, no torn state
    (every returned block has the length the handle announced for it, and after a final reset the stream restarts exactly)."""
    import threading
    rng = np.random.default_rng(31)
    x = (0.2 * rng.standard_normal(512)).astype(np.float32)
    g = capi.Resampler(48000, 16000)
    stop = threading.Event()
    errors = []

    def poll():
        k = 0
        while not stop.is_set():
            try:
                assert g.latency == 59
                g.out_count(512)
                k += 1
                if k % 50 == 0:
                    g.reset()
            except Exception as e:  # pragma: no cover
                errors.append(e)

    t = threading.Thread(target=poll)
    t.start()
    try:
        for _ in range(300):
            y = g.process(x)
            assert len(y) in (151, 170, 171)      # 512 / 3 samples per period, or the first period after a reset (59 held back)
    finally:
        stop.set()
        t.join()
    assert not errors
    g.reset()
    assert np.array_equal(g.process(x), capi.Resampler(48000, 16000).process(x))


def test_upsampling_keeps_every_sample():
    """The documented deviation from rosjack's output stage (include/bfcore.h): with out_rate > in_rate the reference drops
    periods whenever src_process leaves input unconsumed (rosjack.cpp:311-338) and emits at most one block per callback; the
    converter here consumes every input sample and returns the complete conversion -- n_in * ratio outputs, less the look-ahead."""
    g = capi.Resampler(16000, 48000)
    x = (0.2 * np.sin(2 * np.pi * 440.0 * np.arange(20 * 512) / 16000.0)).astype(np.float32)
    y = np.concatenate([g.process(x[i:i + 512]) for i in range(0, len(x), 512)])
    assert abs(len(y) - 3 * (len(x) - g.latency)) <= 3
    # a 440 Hz tone stays a 440 Hz tone at the new rate (first samples: zero history)
    ref = 0.2 * np.sin(2 * np.pi * 440.0 * np.arange(len(y)) / 48000.0)
    assert np.max(np.abs(y[600:] - ref[600:len(y)])) < 2e-3


@pytest.mark.parametrize("rates", [(16000, 48000), (44100, 48000), (48000, 16000), (48000, 44100)])
@pytest.mark.parametrize("period", [512, 1024])
def test_rosjack_stage_mode_matches_the_restatement_of_rosjack(rates, period):
    """BF_RS_ROSJACK against oracle/rosjack_stage_oracle.py (rosjack.cpp:311-349,416-436 around libsamplerate's lazy prepare_data):
    the same periods accepted / dropped, a block published in the same callbacks, the same samples (1e-6 of the scale: the two
    sides differ only by the oracle's accumulated input position); and the complete stream conversion (BF_RS_STREAM) of the SAME
    input beside it -- longer when upsampling, where the reference loses whole periods."""
    from oracle.rosjack_stage_oracle import RosjackStage
    rng = np.random.default_rng(21)
    F = 60
    y = (0.2 * rng.standard_normal(F * period)).astype(np.float32)
    st = RosjackStage(*rates, period)
    g = capi.Resampler(*rates)
    g.set_mode_rosjack(period)
    blocks_o, blocks_g, acc_g, when_o, when_g = [], [], [], [], []
    for t in range(F):
        seg = y[t * period:(t + 1) * period]
        bo = st.callback(seg)
        bg, ag = g.callback(seg)
        acc_g.append(ag)
        if bo is not None:
            blocks_o.append(bo)
            when_o.append(t)
        if bg is not None:
            blocks_g.append(bg)
            when_g.append(t)
    assert acc_g == st.accepted and when_g == when_o and len(blocks_g) > 10
    yo, yg = np.concatenate(blocks_o), np.concatenate(blocks_g)
    assert np.max(np.abs(yo - yg)) <= 1e-6 * np.max(np.abs(yo))
    if rates[1] > rates[0]:
        assert not all(acc_g)                       # the reference drops periods when upsampling ...
        if rates == (16000, 48000):
            assert 0.3 < np.mean(acc_g) < 0.45      # ... about two of three at a ratio of 3
    else:
        assert all(acc_g)
    with pytest.raises(capi.BfError):
        g.process(y[:period])                       # the stream entry refuses a converter that is in rosjack mode
    # the same input through the default mode: every sample converted
    s = capi.Resampler(*rates)
    whole = s.process(y)
    assert abs(len(whole) - int(len(y) * rates[1] / rates[0])) <= 2 * s.latency * max(1, rates[1] // rates[0]) + 2
    if all(acc_g):                                  # nothing dropped: the stage's blocks are a prefix of the stream conversion
        assert np.array_equal(yg, whole[:len(yg)])
    else:
        assert len(whole) > len(yg)
    # back to stream mode on the same object
    g.set_mode_stream()
    assert np.array_equal(g.process(y), whole)


def test_rosjack_stage_on_device_buffers():
    import torch
    period, F = 512, 40
    rng = np.random.default_rng(22)
    y = (0.2 * rng.standard_normal(F * period)).astype(np.float32)
    h = capi.Resampler(16000, 48000)
    h.set_mode_rosjack(period)
    d = capi.Resampler(16000, 48000)
    d.set_mode_rosjack(period)
    yd = torch.from_numpy(y).cuda()
    od = torch.empty(period, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for t in range(F):
        bh, ah = h.callback(y[t * period:(t + 1) * period])
        em, ad = d.callback_device(yd[t * period:].data_ptr(), od.data_ptr(), s)
        assert ad == ah and em == (bh is not None)
        if em:
            torch.cuda.synchronize()
            assert np.array_equal(od.cpu().numpy(), bh)
