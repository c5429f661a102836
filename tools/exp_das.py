#!/usr/bin/env python3
"""A/B of fused-DAS kernel variants on one box (each variant in its own process: the switches are read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import torch
from beamform_amd.capi import Beamformer, BF_INTERLEAVED, BF_PLANAR
from beamform_amd.params import make_params
M, F = int(sys.argv[1]), 65536
p = make_params("das", n_mics=M, theta=20.0)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
y = torch.empty(F * 512, device="cuda")
IL = os.environ.get("EXP_LAYOUT", "planar") == "interleaved"   # same random numbers, read as [sample][mic]
bf = Beamformer(p, layout=BF_INTERLEAVED if IL else BF_PLANAR)
s = torch.cuda.current_stream().cuda_stream
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(8):
        bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
    torch.cuda.synchronize()
bf.reset()
bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
best = 1e9
for rep in range(5):
    ms, msk = bf.time_device(x.data_ptr(), F, y.data_ptr(), 40, s)
    best = min(best, msk)
print(f"kernel {best:.4f} ms  frac {18432*65536/(best*1e-3)/8e12*(M*2048+2048)/18432:.4f}  sha {h}")
''' % ROOT
for M in (8, 4):
    for variant, w64 in [(int(v), t) for t in os.environ.get("EXP_LIBS", ",_prev").split(",") for v in os.environ.get("EXP_VARIANTS", "3").split(",")] * 3:
        env = dict(os.environ, BF_DAS_VARIANT=str(variant), BFCORE_LIB=os.path.join(ROOT, "beamform_amd", "lib", f"libbfcore{w64}.so"))
        out = subprocess.run([sys.executable, "-c", child, str(M)], env=env, capture_output=True, text=True)
        print(f"M={M} variant={variant} w64={w64}: {out.stdout.strip()} {out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
