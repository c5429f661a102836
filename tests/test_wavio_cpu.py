"""File half of the rosjack output stage (rosjack.cpp:189-210, 404-409) and the batch front-end, CPU side: host functions of
the C ABI (no HIP device needed) against the oracle restatement, the golden file and Python's own `wave` module."""
import os
import wave

import numpy as np
import pytest

from beamform_amd import capi
from oracle import wav_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wav_pcm16.npz")


def test_oracle_reproduces_the_golden_file():
    g = np.load(GOLD)
    assert np.array_equal(wav_oracle.float_to_pcm16(g["x"]), g["pcm"])
    assert wav_oracle.wav_bytes(g["x"], int(g["sample_rate"])) == g["wav"].tobytes()
    # known answers of the rule: lrintf(x * 32767), ties to even, no clipping (the cast to short wraps)
    x = np.array([1.0, -1.0, 0.5, 1.5 / 32767, 2.5 / 32767, 32768.0 / 32767.0, 2.0], np.float32)
    assert wav_oracle.float_to_pcm16(x).tolist() == [32767, -32767, 16384, 2, 2, -32768, -2]


def test_host_conversion_and_writer_match_the_oracle(tmp_path):
    g = np.load(GOLD)
    x = g["x"]
    assert np.array_equal(capi.float_to_pcm16(x), g["pcm"])
    path = str(tmp_path / "out.wav")
    with capi.WavWriter(path, int(g["sample_rate"])) as w:
        for a in range(0, len(x), 512):                  # one sf_write_float per callback
            w.write(x[a:a + 512])
    assert open(path, "rb").read() == g["wav"].tobytes()
    with wave.open(path, "rb") as f:                      # an independent reader accepts the container
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 48000, len(x))
        assert np.array_equal(np.frombuffer(f.readframes(len(x)), "<i2"), g["pcm"])


def test_reader_front_end(tmp_path):
    rng = np.random.default_rng(3)
    pcm = rng.integers(-32768, 32768, size=(700, 3)).astype("<i2")      # interleaved frames, 3 channels
    p16 = str(tmp_path / "in16.wav")
    with wave.open(p16, "wb") as f:                                       # written by Python's wave module
        f.setnchannels(3)
        f.setsampwidth(2)
        f.setframerate(44100)
        f.writeframes(pcm.tobytes())
    x, sr = capi.read_wav(p16)
    assert sr == 44100 and x.shape == (3, 700) and x.dtype == np.float32
    assert np.array_equal(x, wav_oracle.pcm16_to_float(pcm.T))            # planar, sf_read_float scaling x / 32768
    # 24-bit PCM and float32 WAVs, and the raw planar float file
    s24 = rng.integers(-(1 << 23), 1 << 23, size=(50, 2))
    b = bytearray()
    for v in s24.reshape(-1):
        b += int(v & 0xFFFFFF).to_bytes(3, "little")
    p24 = str(tmp_path / "in24.wav")
    with wave.open(p24, "wb") as f:
        f.setnchannels(2)
        f.setsampwidth(3)
        f.setframerate(48000)
        f.writeframes(bytes(b))
    x24, _ = capi.read_wav(p24)
    assert np.array_equal(x24, (s24.T.astype(np.float32) / np.float32(8388608.0)))
    import struct
    fl = rng.standard_normal((40, 4)).astype("<f4")
    pf = str(tmp_path / "inf.wav")
    with open(pf, "wb") as f:
        data = fl.tobytes()
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + 16 + 8 + 12 + 8 + len(data)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 3, 4, 16000, 16000 * 16, 16, 32) + b"LIST" + struct.pack("<I", 4) + b"abcd" +
                b"data" + struct.pack("<I", len(data)) + data)
    xf, srf = capi.read_wav(pf)
    assert srf == 16000 and np.array_equal(xf, fl.T)
    raw = rng.standard_normal((5, 1024)).astype(np.float32)
    pr = str(tmp_path / "in.f32")
    raw.tofile(pr)
    assert np.array_equal(capi.read_planar_f32(pr, 5), raw)
    with pytest.raises(capi.BfError):
        capi.read_planar_f32(pr, 7)                                       # size is not a multiple of 7 channels
    with pytest.raises(capi.BfError):
        capi.read_wav(str(tmp_path / "missing.wav"))
    with pytest.raises(capi.BfError):
        capi.read_wav(pr)                                                 # not a RIFF file


def test_round_trip_through_our_own_reader(tmp_path):
    x = (np.random.default_rng(9).standard_normal(5000) * 0.25).astype(np.float32)
    path = str(tmp_path / "rt.wav")
    with capi.WavWriter(path, 48000) as w:
        w.write(x)
    y, sr = capi.read_wav(path)
    assert sr == 48000 and y.shape == (1, 5000)
    assert np.array_equal(y[0], wav_oracle.pcm16_to_float(wav_oracle.float_to_pcm16(x)))
    assert np.abs(y[0] - x).max() <= 1.0 / 32767 * 0.5 + 1e-4 * 1.25 / 32768 + 4e-5
