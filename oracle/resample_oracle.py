"""CPU restatement of the sample-rate converter on the reference's output stage -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path is beamform_amd/csrc/resample.hip.

What it follows: rosjack.cpp:159-184 creates `src_new(SRC_SINC_FASTEST, 1, ..)` with src_ratio = ros_output_sample_rate /
rosjack_sample_rate and rosjack.cpp:311-338 calls `src_process` once per JACK period with end_of_input = 0.  The converter is
libsamplerate (un-vendored, un-pinned; 0.1.9 on Ubuntu 20.04): src_sinc.c `sinc_mono_vari_process` and `calc_output_single`,
restated below in its streaming form (double accumulation of the input position with fmod_one / lrint, 12-bit fixed-point filter
index, double accumulators, linear interpolation of the float coefficient table, zero history, outputs held back until
half_filter_chan_len samples of look-ahead exist).

PARITY UNPINNED: libsamplerate's SINC_FASTEST table (fastest_coeffs.h) is not in this image and the reference has no test
vectors for this stage; `default_table()` is the same Kaiser-windowed sinc the library builds (bf_resampler_default_table
returns the library's copy for the comparison).  With libsamplerate's table passed as `coeffs` this is libsamplerate's arithmetic.
"""
from __future__ import annotations

import math

import numpy as np

SHIFT_BITS = 12
FP_ONE = float(1 << SHIFT_BITS)
DEFAULT_LEN = 2464
DEFAULT_INC = 128


def _i0(x: float) -> float:
    s = t = 1.0
    for k in range(1, 200):
        t *= (x / (2.0 * k)) ** 2
        s += t
        if t < 1e-18 * s:
            break
    return s


def default_table(n: int = DEFAULT_LEN, inc: int = DEFAULT_INC) -> np.ndarray:
    fc = 0.83147237295484055508
    beta = 0.1102 * (97.0 - 8.7)
    i0b = _i0(beta)
    t = np.zeros(n, dtype=np.float32)
    for i in range(n):
        x = i / inc * fc
        s = 1.0 if i == 0 else math.sin(math.pi * x) / (math.pi * x)
        u = i / (n - 1)
        w = _i0(beta * math.sqrt(1.0 - u * u)) / i0b if u < 1.0 else 0.0
        t[i] = np.float32(fc * s * w)
    t[n - 1] = 0.0
    return t


def _lrint(x: float) -> int:
    return int(np.rint(x))  # round-half-even, as lrint in the default rounding mode


class SincResampler:
    """One mono converter: process(block) -> the output samples libsamplerate's src_process would have generated so far."""

    def __init__(self, in_rate: int, out_rate: int, coeffs: np.ndarray | None = None, index_inc: int = DEFAULT_INC):
        self.ratio = float(out_rate) / float(in_rate)
        self.coeffs = np.asarray(default_table() if coeffs is None else coeffs, dtype=np.float32).astype(np.float64)
        self.index_inc = int(index_inc)
        self.coeff_half_len = len(self.coeffs) - 2
        count = (self.coeff_half_len + 2.0) / self.index_inc
        if self.ratio < 1.0:
            count /= self.ratio
        self.half_len = _lrint(count) + 1            # half_filter_chan_len for one channel
        self.buf = np.zeros(0, dtype=np.float64)     # every input sample so far (a linear buffer instead of libsamplerate's ring)
        self.b_current = 0                           # index of the input sample at the integer part of the position
        self.input_index = 0.0                       # fractional part (src_sinc.c keeps it in psrc->last_position)

    def _calc_output_single(self, increment: int, start_filter_index: int) -> float:
        max_filter_index = self.coeff_half_len << SHIFT_BITS
        c, buf = self.coeffs, self.buf

        def wing(filter_index, data_index, step, stop_at_zero_inclusive):
            acc = 0.0
            while True:
                fraction = (filter_index & ((1 << SHIFT_BITS) - 1)) / FP_ONE
                indx = filter_index >> SHIFT_BITS
                icoeff = c[indx] + fraction * (c[indx + 1] - c[indx])
                x = buf[data_index] if 0 <= data_index < len(buf) else 0.0
                acc += icoeff * x
                filter_index -= increment
                data_index += step
                if stop_at_zero_inclusive:
                    if filter_index < 0:
                        break
                elif filter_index <= 0:
                    break
            return acc

        filter_index = start_filter_index
        coeff_count = (max_filter_index - filter_index) // increment
        filter_index += coeff_count * increment
        left = wing(filter_index, self.b_current - coeff_count, +1, True)
        filter_index = increment - start_filter_index
        coeff_count = (max_filter_index - filter_index) // increment
        filter_index += coeff_count * increment
        right = wing(filter_index, self.b_current + 1 + coeff_count, -1, False)
        return left + right

    def process(self, block) -> np.ndarray:
        self.buf = np.concatenate([self.buf, np.asarray(block, dtype=np.float32).astype(np.float64)])
        out = []
        src_ratio = self.ratio
        while len(self.buf) - self.b_current > self.half_len:      # samples_in_hand > half_filter_chan_len
            float_increment = self.index_inc * (src_ratio if src_ratio < 1.0 else 1.0)
            increment = _lrint(float_increment * FP_ONE)
            start_filter_index = _lrint(self.input_index * float_increment * FP_ONE)
            v = (float_increment / self.index_inc) * self._calc_output_single(increment, start_filter_index)
            out.append(np.float32(v))
            self.input_index += 1.0 / src_ratio
            rem = math.fmod(self.input_index, 1.0)                 # fmod_one (positions are never negative here)
            self.b_current += _lrint(self.input_index - rem)
            self.input_index = rem
        return np.asarray(out, dtype=np.float32)


def resample_vectorised(x, in_rate: int, out_rate: int, coeffs: np.ndarray | None = None, index_inc: int = DEFAULT_INC) -> np.ndarray:
    """The same filter evaluated with numpy at the exact rational positions K * in_rate / out_rate (what the HIP kernel computes);
    used for the larger test sizes.  Differs from SincResampler only through libsamplerate's accumulated position (~1e-16 per step)."""
    x = np.asarray(x, dtype=np.float32).astype(np.float64)
    c = np.asarray(default_table() if coeffs is None else coeffs, dtype=np.float32).astype(np.float64)
    ratio = out_rate / in_rate
    half = len(c) - 2
    count = (half + 2.0) / index_inc / (ratio if ratio < 1.0 else 1.0)
    H = _lrint(count) + 1
    lim = len(x) - H
    if lim <= 0:
        return np.zeros(0, dtype=np.float32)
    n_out = (lim * out_rate - 1) // in_rate + 1
    K = np.arange(n_out, dtype=np.int64)
    num = K * in_rate
    cur = num // out_rate
    frac = (num % out_rate) / float(out_rate)
    float_increment = index_inc * (ratio if ratio < 1.0 else 1.0)
    increment = _lrint(float_increment * FP_ONE)
    start = np.rint(frac * float_increment * FP_ONE).astype(np.int64)
    max_fi = half << SHIFT_BITS
    xp = np.concatenate([np.zeros(H + 2), x, np.zeros(H + 2)])
    off = H + 2
    acc = np.zeros(n_out)
    # left wing: tap j = 0 is cur, j = 1 is cur - 1, ...: filter index start + j * increment
    j = 0
    while True:
        fi = start + j * increment
        live = fi <= max_fi
        if not live.any():
            break
        fi = np.where(live, fi, 0)
        ic = c[fi >> SHIFT_BITS] + (fi & ((1 << SHIFT_BITS) - 1)) / FP_ONE * (c[(fi >> SHIFT_BITS) + 1] - c[fi >> SHIFT_BITS])
        acc += np.where(live, ic * xp[cur - j + off], 0.0)
        j += 1
    j = 0
    while True:
        fi = increment - start + j * increment
        live = (fi <= max_fi) & (fi > 0)
        if not (fi <= max_fi).any():
            break
        fi = np.where(live, fi, 0)
        ic = c[fi >> SHIFT_BITS] + (fi & ((1 << SHIFT_BITS) - 1)) / FP_ONE * (c[(fi >> SHIFT_BITS) + 1] - c[fi >> SHIFT_BITS])
        acc += np.where(live, ic * xp[cur + 1 + j + off], 0.0)
        j += 1
    return ((float_increment / index_inc) * acc).astype(np.float32)
