#!/usr/bin/env python3
"""Time das at the reference's precision on the headline batch: tools/time_das_f64.py [mics] [frames]  (BF_DAS_F64_SCHED selects the chunk plan; BFCORE_LIB an A/B build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd.capi import BF_DAS_F64, Beamformer
from beamform_amd.params import make_params
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
F = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
p = make_params("das", n_mics=M)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
y = torch.empty(F * 512, device="cuda")
bf = Beamformer(p, das_impl=BF_DAS_F64)
s = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
tk = [bf.time_device(x.data_ptr(), F, y.data_ptr(), 10, s) for _ in range(5)]
ts = [t[0] for t in tk]
yy = y.cpu().numpy()
print(f"kernel-only (event pair around the launch): best {min(t[1] for t in tk):.4f} ms")
print(f"das f64 (BF_DAS_F64_SCHED={os.environ.get('BF_DAS_F64_SCHED', 'default')}) {M}-mic {F} frames: best {min(ts):.4f} ms, median {sorted(ts)[2]:.4f} ms; "
      f"{18432 * F / (min(ts) * 1e-3) / 8e12 * (M * 2048 + 2048) / 18432:.4f} of 8 TB/s; checksum {float(np.abs(yy).sum()):.6f}", flush=True)
