#!/usr/bin/env python3
"""Experiment: do independent bin-pipeline chains on separate HIP streams overlap (stft HBM-bound vs per-bin VALU-bound)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
algo = sys.argv[1] if len(sys.argv) > 1 else "mvdr"
F = 65536
p = make_params(algo, n_mics=8)
x = torch.rand((8, F * 512), device="cuda") - 0.5
y = torch.empty(F * 512, device="cuda")
for n in (1, 2, 3, 4, 8):
    bfs = [Beamformer(p) for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    per = F // n
    xs = [x[:, i * per * 512:(i + 1) * per * 512].contiguous() for i in range(n)]
    def go():
        for i in range(n):
            bfs[i].process_device(xs[i].data_ptr(), per, y[i * per * 512:].data_ptr(), 0, streams[i].cuda_stream)
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        go()
    torch.cuda.synchronize()
    print(f"{algo}: {n} concurrent chains of {per} frames: {(time.perf_counter() - t0) * 100:.3f} ms per 65536 frames")
    for b in bfs:
        b.close()
