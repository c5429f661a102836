#!/usr/bin/env python3
"""Race hunt for the one-launch fp64 das kernels: the same batch many times against the fused fp32 kernel's output (differences
beyond float rounding = a hop completed from a stale partner).  tools/stress_das_f64.py [reps] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd.capi import BF_DAS_F64, Beamformer
from beamform_amd.params import make_params
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
F = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
M = 8
p = make_params("das", n_mics=M, theta=35.0)
g = torch.Generator(device="cuda").manual_seed(21)
x = torch.rand(M, F * 512, device="cuda", generator=g) - 0.5
y = torch.empty(F * 512, device="cuda")
y32 = torch.empty(F * 512, device="cuda")
from beamform_amd.capi import BF_DAS_FUSED_F32
Beamformer(p, das_impl=BF_DAS_FUSED_F32).process_device(x.data_ptr(), F, y32.data_ptr())
torch.cuda.synchronize()
n_bad = 0
for rep in range(reps):
    bf = Beamformer(p, das_impl=BF_DAS_F64)
    y.fill_(float("nan"))
    bf.process_device(x.data_ptr(), F, y.data_ptr())
    torch.cuda.synchronize()
    d = (y - y32).view(F, 512).abs().max(dim=1).values
    bad = torch.nonzero(~(d < 1e-4)).flatten().cpu().numpy()
    if len(bad):
        n_bad += 1
        print(f"rep {rep}: {len(bad)} bad hops, first {bad[:16]} (mod 16: {sorted(set(int(b) % 16 for b in bad))})", flush=True)
    bf.close()
print(f"{reps} repetitions of {F} frames: {n_bad} with bad hops")
