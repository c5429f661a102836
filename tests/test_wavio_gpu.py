"""Output stage on the GPU box: the device float -> PCM16 conversion of a resident batch, and a node driven from a WAV
file that writes the WAV file rosjack's write_file option would (rosjack.cpp:189-210, 404-409)."""
import os
import subprocess
import wave

import numpy as np
import pytest

from beamform_amd import capi
from beamform_amd.params import AIRA16_XY, make_params
from beamform_amd.synth import make_scene
from conftest import ROOT
from oracle import wav_oracle

pytestmark = pytest.mark.gpu


def test_device_conversion_is_the_host_rule():
    import torch
    assert torch.cuda.is_available()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "wav_pcm16.npz"))
    rng = np.random.default_rng(1)
    x = np.concatenate([g["x"], (rng.standard_normal(1 << 20) * 0.4).astype(np.float32), np.float32([1.7, -3.2, 0.3])])
    xd = torch.from_numpy(x).cuda()
    od = torch.empty(len(x), dtype=torch.int16, device="cuda")
    capi.float_to_pcm16_device(xd.data_ptr(), od.data_ptr(), len(x))
    torch.cuda.synchronize()
    assert np.array_equal(od.cpu().numpy(), wav_oracle.float_to_pcm16(x))
    assert np.array_equal(od.cpu().numpy(), capi.float_to_pcm16(x))


def test_wav_in_wav_out_node(tmp_path):
    import oracle
    exe = os.path.join(ROOT, "examples", "file_node")
    M, F = 4, 20
    lines = ["initial_angle: 15.0"] + [f"mic{i}: {{id: {i}, x: {x:.3f}, y: {y:.3f}, z: 0.000}}" for i, (x, y) in enumerate(AIRA16_XY[:M])]
    cfg = tmp_path / "beamform_config.yaml"
    cfg.write_text("\n".join(lines) + "\n")
    x = make_scene(M, F, seed=66)
    pcm_in = wav_oracle.float_to_pcm16(x)                      # a 4-channel PCM16 recording
    with wave.open(str(tmp_path / "in.wav"), "wb") as f:
        f.setnchannels(M)
        f.setsampwidth(2)
        f.setframerate(48000)
        f.writeframes(np.ascontiguousarray(pcm_in.T).astype("<i2").tobytes())
    subprocess.check_call([exe, "das", str(cfg), str(tmp_path / "in.wav"), str(tmp_path / "out.wav")])
    x_q = wav_oracle.pcm16_to_float(pcm_in)                    # what sf_read_float hands the node
    y_ref, _ = oracle.OracleNode(make_params("das", n_mics=M, theta=15.0)).process(x_q)
    want = wav_oracle.float_to_pcm16(y_ref)
    with wave.open(str(tmp_path / "out.wav"), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 48000, F * 512)
        got = np.frombuffer(f.readframes(F * 512), "<i2")
    # the node's float output is within 1e-5 of the reference's; one PCM step is 3e-5: almost every sample is identical
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.99


def test_wav_out_at_ros_output_sample_rate(tmp_path):
    """file_node with out_rate = 16000: every period goes through the converter (rosjack.cpp:410-427) and the WAV header carries
    the resampled rate (rosjack.cpp:192-195)."""
    import oracle
    from oracle.resample_oracle import SincResampler
    exe = os.path.join(ROOT, "examples", "file_node")
    M, F = 4, 24
    lines = ["initial_angle: 15.0"] + [f"mic{i}: {{id: {i}, x: {x:.3f}, y: {y:.3f}, z: 0.000}}" for i, (x, y) in enumerate(AIRA16_XY[:M])]
    cfg = tmp_path / "beamform_config.yaml"
    cfg.write_text("\n".join(lines) + "\n")
    x = make_scene(M, F, seed=67)
    x.astype("<f4").tofile(str(tmp_path / "in.f32"))
    subprocess.check_call([exe, "das", str(cfg), str(tmp_path / "in.f32"), str(tmp_path / "out.wav"), "-", "16000"])
    y_ref, _ = oracle.OracleNode(make_params("das", n_mics=M, theta=15.0)).process(x)
    rs = SincResampler(48000, 16000)
    want = wav_oracle.float_to_pcm16(np.concatenate([rs.process(y_ref[i:i + 512].astype(np.float32)) for i in range(0, F * 512, 512)]))
    with wave.open(str(tmp_path / "out.wav"), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, 16000)
        got = np.frombuffer(f.readframes(f.getnframes()), "<i2")
    assert len(got) == len(want) == (F * 512 - rs.half_len + 2) // 3
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.98
