#!/usr/bin/env python3
"""mvdr / lcmv over frequency bands: the default band, a band that starts at 0 Hz and the full band (both reach the irregular
problems of quirk Q1 and take the group kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer, BF_PLANAR
from beamform_amd.params import make_params
F = 65536
for algo, M, interf in (("mvdr", 8, ()), ("lcmv", 8, (-60.0, 90.0)), ("mvdr", 3, ())):
    for band in ((100.0, 16000.0), (0.0, 16000.0), (0.0, 24000.0), (100.0, 24000.0)):
        p = make_params(algo, n_mics=M, interf=interf, freq_min=band[0], freq_max=band[1])
        bf = Beamformer(p, n_streams=1, layout=BF_PLANAR)
        x = torch.rand((1, M, F * 512), device="cuda") - 0.5
        y = torch.empty((1, F * 512), device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
        torch.cuda.synchronize()
        ms, _ = bf.time_device(x.data_ptr(), F, y.data_ptr(), 5, s)
        print(f"{algo} M={M} band {band[0]:.0f}-{band[1]:.0f} Hz: {ms:.3f} ms", flush=True)
