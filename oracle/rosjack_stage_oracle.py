"""CPU restatement of rosjack's resampling output stage as the reference RUNS it -- TEST INFRASTRUCTURE ONLY.

oracle/resample_oracle.py restates the converter (libsamplerate's sinc_mono_vari_process as a stream operator).  This module
restates what rosjack does AROUND it, callback by callback (rosjack.cpp):

  :176-183  samplerate_data.data_out has room for rosjack_window_size floats and output_frames = rosjack_window_size: one
            src_process call never returns more than one period of output;
  :311-338  convert_to_sample_rate: the new period is copied into the converter's input ONLY when input_frames == 0 -- while
            src_process has left input unconsumed, the callback's period is silently DROPPED; one src_process per callback; what
            it generated goes to a ring, data_in / input_frames advance by input_frames_used;
  :340-349  convert_to_sample_rate_ready + :416-436: a block of data_length samples is published / written only when the ring
            holds that many; at most one block per callback; nothing flushes the tail.

and the part of src_process that decides input_frames_used (libsamplerate 0.1.9, src_sinc.c: sinc_set_converter's b_len,
prepare_data, the reload test at the top of sinc_mono_vari_process' loop): input is pulled into the converter's buffer only when
fewer than half_filter_chan_len + 1 samples are in hand, and then as much as fits behind b_end.  When the output rate is the
higher one a call stops at its output_frames with input left over, the next call needs no reload, input_frames_used is 0, and
the period after that is dropped: 16 kHz -> 48 kHz keeps roughly every third period, 44.1 -> 48 kHz drops about one in twelve.
For out_rate <= in_rate every period is consumed in its own callback and only the ring's block rule differs from the stream.

PARITY UNPINNED: libsamplerate is not in the image (un-vendored dependency); b_len and prepare_data are restated from its 0.1.9
source as published.  Only tests/ may import this module.
"""
from __future__ import annotations

import math

import numpy as np

from . import resample_oracle as ro

SRC_MAX_RATIO = 256


class RosjackStage:
    def __init__(self, in_rate: int, out_rate: int, period: int, coeffs=None, index_inc: int = ro.DEFAULT_INC):
        self.period = int(period)
        self.conv = ro.SincResampler(in_rate, out_rate, coeffs, index_inc)   # values; its buf = every sample the converter has loaded
        c = self.conv
        # sinc_set_converter (0.1.9): b_len = max(lrint(2.5 * coeff_half_len / index_inc * SRC_MAX_RATIO), 4096) * channels
        self.b_len = max(ro._lrint(2.5 * c.coeff_half_len / (c.index_inc * 1.0) * SRC_MAX_RATIO), 4096)
        self.b_current = 0          # buffer-relative, as libsamplerate keeps them: only their distances to b_len matter here
        self.b_end = 0
        self.pending = np.zeros(0, np.float32)   # samplerate_buff_in from data_in on: input_frames samples not yet pulled in
        self.ring = np.zeros(0, np.float32)      # samplerate_circbuff between the read and the write index
        self.accepted = []                       # per callback: was the period copied in (True) or dropped (False)
        self.used = []                           # per callback: input_frames_used

    def _prepare_data(self) -> int:
        """prepare_data: how many pending samples the converter pulls in now (and the buffer bookkeeping that decides it)."""
        half = self.conv.half_len
        if self.b_current == 0:     # initial state: zeros in front, then data
            n = self.b_len - 2 * half
            self.b_current = self.b_end = half
        elif self.b_end + half + 1 < self.b_len:
            n = max(self.b_len - self.b_current - half, 0)
        else:                       # move what is left to the start of the buffer
            keep = self.b_end - self.b_current
            self.b_current = half
            self.b_end = self.b_current + keep
            n = max(self.b_len - self.b_current - half, 0)
        n = min(len(self.pending), n)
        self.b_end += n
        return n

    def _src_process(self) -> np.ndarray:
        c = self.conv
        out, used = [], 0
        while len(out) < self.period:                                  # out_gen < out_count (= output_frames = the period)
            if self.b_end - self.b_current <= c.half_len:              # samples_in_hand <= half_filter_chan_len: reload
                n = self._prepare_data()
                c.buf = np.concatenate([c.buf, self.pending[:n].astype(np.float64)])
                self.pending = self.pending[n:]
                used += n
                if self.b_end - self.b_current <= c.half_len:
                    break
            float_increment = c.index_inc * (c.ratio if c.ratio < 1.0 else 1.0)
            increment = ro._lrint(float_increment * ro.FP_ONE)
            start_filter_index = ro._lrint(c.input_index * float_increment * ro.FP_ONE)
            out.append(np.float32((float_increment / c.index_inc) * c._calc_output_single(increment, start_filter_index)))
            c.input_index += 1.0 / c.ratio
            rem = math.fmod(c.input_index, 1.0)
            step = ro._lrint(c.input_index - rem)
            c.b_current += step                                        # absolute position in c.buf
            self.b_current += step                                     # the same step in the converter's own buffer coordinates
            c.input_index = rem
        self.used.append(used)
        return np.asarray(out, np.float32)

    def callback(self, data_out):
        """One output_to_rosjack(data_out, data_length = period): returns the published block (period samples) or None."""
        data_out = np.asarray(data_out, np.float32)
        assert len(data_out) == self.period
        if len(self.pending) == 0:                                     # rosjack.cpp:314-320
            self.pending = data_out.copy()
            self.accepted.append(True)
        else:
            self.accepted.append(False)                                # the period never reaches the converter
        self.ring = np.concatenate([self.ring, self._src_process()])   # :323-332
        if len(self.ring) >= self.period:                              # :340-349, :416-436
            block, self.ring = self.ring[:self.period], self.ring[self.period:]
            return block
        return None

    def run(self, y):
        """Feed a whole output stream period by period; returns (concatenated published blocks, accepted flags)."""
        y = np.asarray(y, np.float32)
        blocks = [b for t in range(len(y) // self.period) if (b := self.callback(y[t * self.period:(t + 1) * self.period])) is not None]
        return (np.concatenate(blocks) if blocks else np.zeros(0, np.float32)), np.array(self.accepted)
