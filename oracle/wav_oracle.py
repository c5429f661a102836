"""numpy restatement of the file half of the rosjack output stage (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Reference call sites: rosjack.cpp:189-210 (sf_open: SF_FORMAT_WAV | SF_FORMAT_PCM_16, channels = 1) and rosjack.cpp:404-409
(sf_write_float of every output period).  The arithmetic lives in libsndfile (a system dependency of the reference --
Ubuntu 20.04: 1.0.28 -- absent from /root/reference and from this image), restated from its published behaviour:
  * float -> PCM16 with norm_float = SF_TRUE (default) and clipping off (default): src/pcm.c f2s_array,
        dest[i] = lrintf(src[i] * (1.0 * 0x7FFF))   stored as short
    i.e. a float32 product, round-half-to-even, and the C cast to short (modulo 2^16: NO saturation);
  * PCM16 -> float on read (sf_read_float, s2f_array): src[i] * (1.0 / 0x8000);
  * header of a WAVE_FORMAT_PCM file: 'RIFF' <36 + data bytes> 'WAVE' 'fmt ' <16> <1> <channels> <rate> <rate * block>
    <block> <bits> 'data' <data bytes>  (44 bytes, little-endian).
PARITY UNPINNED against libsndfile itself (not installed); pinned against Python's own `wave` module for the container
format (tests/test_wavio_cpu.py) and against the golden file tests/golden/wav_pcm16.npz."""
from __future__ import annotations

import struct

import numpy as np


def float_to_pcm16(x) -> np.ndarray:
    scaled = (np.asarray(x, np.float32) * np.float32(32767.0)).astype(np.float32)   # float product
    r = np.rint(scaled.astype(np.float64)).astype(np.int64)                          # lrintf: nearest-even (exact in double)
    return (r & 0xFFFF).astype(np.uint16).view(np.int16)                             # (short): modulo 2^16


def pcm16_to_float(s) -> np.ndarray:
    return (np.asarray(s, np.int16).astype(np.float32) * np.float32(1.0 / 32768.0)).astype(np.float32)


def wav_bytes(x, sample_rate: int) -> bytes:
    pcm = float_to_pcm16(x).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16)
    return hdr + b"data" + struct.pack("<I", len(pcm)) + pcm
