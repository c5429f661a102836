#!/usr/bin/env python3
"""Latency of the streaming entry point bf_process_hop (one jack_callback worth of work, host buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
for algo in ("das", "mvdr", "phasempf"):
    p = make_params(algo, n_mics=8)
    bf = Beamformer(p)
    x = (np.random.default_rng(0).random((8, 512), dtype=np.float32) - 0.5)
    for _ in range(20):
        bf.process_hop(x)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        bf.process_hop(x)
    dt = (time.perf_counter() - t0) / n
    print(f"{algo}: {dt*1e6:.0f} us per hop (JACK period at 48 kHz: 10667 us)")
