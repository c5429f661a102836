#!/usr/bin/env python3
"""Latency of the streaming entry point bf_process_hop (one jack_callback worth of work, host buffers) at every JACK period."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
hops = [int(h) for h in sys.argv[1:]] or [64, 128, 256, 512, 1024]
for hop in hops:
    for algo in ("das", "mvdr", "phase", "phasempf", "lcmv"):
        p = make_params(algo, n_mics=8, hop=hop, interf=(-60.0, 90.0) if algo == "lcmv" else ())
        bf = Beamformer(p)
        x = (np.random.default_rng(0).random((8, hop), dtype=np.float32) - 0.5)
        for _ in range(30):
            bf.process_hop(x)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            bf.process_hop(x)
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts)) * 1e6
        print(f"period {hop:5d} {algo:9s}: median {ts[len(ts) // 2]:6.0f} us, p99 {ts[int(0.99 * len(ts))]:6.0f} us per callback (JACK budget at 48 kHz: {hop / 48000 * 1e6:6.0f} us)", flush=True)
