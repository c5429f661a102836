"""Writes tests/golden/wav_pcm16.npz: input samples (edge cases of the float -> short rule included) and the bytes of the
WAV file the oracle restatement (oracle/wav_oracle.py) produces for them.  Run from the repo root."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from oracle import wav_oracle  # noqa: E402

rng = np.random.default_rng(2024)
x = np.concatenate([
    np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.0 / 32767, 1.5 / 32767, 2.5 / 32767, -1.5 / 32767, -2.5 / 32767,   # ties: even
              0.99999, 1.00002, 1.2, -1.00004, -1.5, 2.0, 3.0e-5, 1e-9, 32768.0 / 32767.0], np.float32),
    (rng.standard_normal(2000) * 0.3).astype(np.float32),
])
np.savez_compressed(os.path.join("tests", "golden", "wav_pcm16.npz"), x=x, pcm=wav_oracle.float_to_pcm16(x),
                    wav=np.frombuffer(wav_oracle.wav_bytes(x, 48000), np.uint8), sample_rate=48000)
print("wrote", len(x), "samples")
