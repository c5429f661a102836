#!/usr/bin/env python3
"""Time every node over microphone counts / interferer counts (one process, 16 384-frame batches of noise): a table to spot shapes
that fall onto a slow kernel.  ms per batch and microseconds per 1 000 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer, BF_PLANAR
from beamform_amd.params import make_params

F = int(os.environ.get("SWEEP_FRAMES", "16384"))
cases = []
for M in (2, 3, 4, 5, 6, 8, 9, 12, 16):  # make_params lays out up to 16 microphones by itself
    cases.append(("das", M, 0))
    cases.append(("mvdr", M, 0))
    cases.append(("phase", M, 0))
    for K in (1, 2, 3):
        if K + 1 <= M:
            cases.append(("lcmv", M, K))
    # gss / phasempf / gsc recurse over frames or samples of a stream: one long stream says nothing about them
angles = (-60.0, 90.0, 150.0)
for algo, M, K in cases:
    try:
        p = make_params(algo, n_mics=M, interf=angles[:K])
        bf = Beamformer(p, n_streams=1, layout=BF_PLANAR)
        x = torch.rand((1, M, F * 512), device="cuda") - 0.5
        y = torch.empty((1, F * 512), device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        t0 = time.perf_counter()
        n = 0
        while n < 3 or time.perf_counter() - t0 < 0.05:
            bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
            n += 1
        torch.cuda.synchronize()
        ms, _ = bf.time_device(x.data_ptr(), F, y.data_ptr(), 5, s)
        print(f"{algo:9s} M={M:2d} K={K}: {ms:8.3f} ms  {ms * 1e3 / (F / 1000):8.2f} us/kframe", flush=True)
        del bf, x, y
    except Exception as e:  # noqa
        print(f"{algo:9s} M={M:2d} K={K}: {type(e).__name__}: {e}", flush=True)
