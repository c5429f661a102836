#!/usr/bin/env python3
"""lcmv over microphone / interferer counts, 65 536-frame batches of noise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer, BF_PLANAR
from beamform_amd.params import make_params
F = 65536
angles = (-60.0, 90.0, 150.0)
for M, K in ((2, 1), (3, 0), (3, 1), (3, 2), (4, 2), (4, 3), (6, 2), (6, 3), (8, 1), (8, 2), (8, 3)):
    p = make_params("lcmv", n_mics=M, interf=angles[:K])
    bf = Beamformer(p, n_streams=1, layout=BF_PLANAR)
    x = torch.rand((1, M, F * 512), device="cuda") - 0.5
    y = torch.empty((1, F * 512), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
    torch.cuda.synchronize()
    ms, _ = bf.time_device(x.data_ptr(), F, y.data_ptr(), 10, s)
    print(f"lcmv M={M} K={K}: {ms:.3f} ms", flush=True)
