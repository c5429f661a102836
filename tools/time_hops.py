#!/usr/bin/env python3
"""Nodes at the three JACK periods, the same number of samples (the headline batch's): per-call time and the kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer, BF_PLANAR
from beamform_amd.params import make_params
NS = 65536 * 512
for algo in ("mvdr", "phase", "lcmv"):
    for hop in (64, 256, 512, 1024):
        interf = (-60.0, 90.0) if algo == "lcmv" else ()
        p = make_params(algo, n_mics=8, hop=hop, interf=interf)
        F = NS // hop
        bf = Beamformer(p, n_streams=1, layout=BF_PLANAR)
        x = torch.rand((1, 8, NS), device="cuda") - 0.5
        y = torch.empty((1, NS), device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
        torch.cuda.synchronize()
        ms, _ = bf.time_device(x.data_ptr(), F, y.data_ptr(), 5, s)
        print(f"{algo} period {hop}: {ms:.3f} ms", flush=True)
