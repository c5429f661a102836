"""Sample-rate converter of the output stage (rosjack.cpp:159-184, 311-338): the oracle's two forms agree, and the filter
behaves like the band-limited interpolator libsamplerate documents for SRC_SINC_FASTEST.  No GPU."""
import numpy as np
import pytest

from oracle.resample_oracle import DEFAULT_INC, SincResampler, default_table, resample_vectorised


@pytest.mark.parametrize("rates", [(48000, 16000), (48000, 44100), (44100, 48000), (16000, 48000), (48000, 48000), (48000, 8000)])
def test_streaming_restatement_equals_vectorised(rates):
    rng = np.random.default_rng(3)
    x = (0.2 * rng.standard_normal(5 * 512)).astype(np.float32)
    r = SincResampler(*rates)
    y = np.concatenate([r.process(x[i:i + 512]) for i in range(0, len(x), 512)])   # one src_process per JACK period
    v = resample_vectorised(x, *rates)
    assert len(y) == len(v) > 0
    # libsamplerate accumulates the position in a double; the exact rational position differs by ~1e-16 per step
    np.testing.assert_allclose(y, v, rtol=0, atol=2e-7)


def test_block_size_does_not_matter():
    rng = np.random.default_rng(4)
    x = (0.2 * rng.standard_normal(4000)).astype(np.float32)
    a = SincResampler(48000, 16000)
    ya = np.concatenate([a.process(x[i:i + 512]) for i in range(0, len(x), 512)])
    b = SincResampler(48000, 16000)
    yb = np.concatenate([b.process(x[i:i + 77]) for i in range(0, len(x), 77)])
    n = min(len(ya), len(yb))
    assert abs(len(ya) - len(yb)) <= 1 and np.array_equal(ya[:n], yb[:n])


def test_output_count_and_latency():
    r = SincResampler(48000, 16000)
    assert r.half_len == 59                     # lrint(2464 / 128 * 3) + 1
    assert len(r.process(np.zeros(59, np.float32))) == 0
    assert len(r.process(np.zeros(1, np.float32))) == 1     # the first output needs cur + half_len < available
    r1 = SincResampler(48000, 48000)
    assert r1.half_len == 20


@pytest.mark.parametrize("f0,lo,hi", [(1000.0, 0.9999, 1.0001), (5000.0, 0.999, 1.001), (12000.0, 0.0, 2e-5)])
def test_sine_response_48k_to_16k(f0, lo, hi):
    n = np.arange(24000)
    x = np.sin(2 * np.pi * f0 * n / 48000).astype(np.float32)
    y = resample_vectorised(x, 48000, 16000)[300:-300]
    gain = np.sqrt(2 * np.mean(y.astype(np.float64) ** 2))
    assert lo <= gain <= hi                     # pass band flat, 12 kHz (would alias to 4 kHz) down by > 94 dB
    if f0 < 6000:
        m = np.arange(300, 300 + len(y))
        err = y - np.sin(2 * np.pi * f0 * m / 16000)
        assert np.sqrt(np.mean(err ** 2)) < 2e-4 * (f0 / 1000.0)


def test_table_geometry():
    t = default_table()
    assert len(t) == 2464 and DEFAULT_INC == 128 and t[-1] == 0.0
    assert abs(float(t[0]) - 0.8314723730) < 1e-7      # libsamplerate's published first SINC_FASTEST coefficient
    assert abs(float(t[1]) - 0.8314140055) < 1e-7      # ... and the second


def test_rosjack_stage_restatement_drops_periods_when_upsampling():
    """oracle/rosjack_stage_oracle.py (rosjack.cpp:311-349,416-436 around libsamplerate's lazy prepare_data): what it publishes is
    the stream conversion of the periods it accepted, one block per callback; 16 -> 48 kHz keeps roughly every third period."""
    from oracle.rosjack_stage_oracle import RosjackStage
    from oracle.resample_oracle import SincResampler
    rng = np.random.default_rng(3)
    P, F = 256, 24
    y = (0.1 * rng.standard_normal(F * P)).astype(np.float32)
    for rates, frac in (((16000, 48000), (0.3, 0.5)), ((44100, 48000), (0.85, 0.99)), ((48000, 16000), (1.0, 1.0))):
        st = RosjackStage(*rates, P)
        out, acc = st.run(y)
        assert frac[0] <= acc.mean() <= frac[1], (rates, acc.mean())
        kept = np.concatenate([y[t * P:(t + 1) * P] for t in range(F) if acc[t]])
        ref = SincResampler(*rates).process(kept)
        assert len(out) % P == 0 and len(out) > 0 and np.array_equal(out, ref[:len(out)])
        assert all(u in (0, P) for u in st.used)          # libsamplerate's buffer (12 310 samples) always has room for a whole period
    st = RosjackStage(16000, 48000, P)
    st.run(y)
    assert st.accepted[:9] == [True, True, False, True, False, False, True, False, False]
