#!/usr/bin/env python3
"""Time one node on a 65 536-frame batch of uniform noise: tools/time_node.py <algo> [mics] [frames] (env switches apply)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
algo = sys.argv[1] if len(sys.argv) > 1 else "mvdr"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
interf = (-60.0, 90.0, 150.0) if algo in ("lcmv", "gss") else ()
p = make_params(algo, n_mics=M, interf=interf)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
y = torch.empty(F * 512, device="cuda")
bf = Beamformer(p)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
best = min(bf.time_device(x.data_ptr(), F, y.data_ptr(), 5, s)[0] for _ in range(3))
yy = y.cpu().numpy()
print(f"{algo} {M}-mic {F} frames: step {best:.3f} ms  checksum {float(np.nansum(np.abs(yy[5120:]))):.6f}")
