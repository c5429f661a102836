#!/usr/bin/env python3
"""Minimal driver for profiling: N launches of the batch path on resident data (no CPU baseline)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import Beamformer, BF_PLANAR, BF_INTERLEAVED
from beamform_amd.params import make_params

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="das")
ap.add_argument("--mics", type=int, default=8)
ap.add_argument("--frames", type=int, default=65536)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--layout", default="planar")
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--settle-ms", type=float, default=80.0)
ap.add_argument("--dirs", type=int, default=1, help="look directions per input stream (same input, n_dirs outputs)")
ap.add_argument("--das-f64", action="store_true", help="das in double (BF_DAS_F64, the library default); without it this tool runs the fused fp32 opt-in")
ap.add_argument("--mixed", action="store_true", help="bf_config.precision = BF_PRECISION_MIXED (z48 spectra, fp32 backward transform)")
ap.add_argument("--hop", type=int, default=512, help="JACK period (frames are 2 * hop samples); --frames counts frames of that period")
a = ap.parse_args()
interf = (-60.0, 90.0, 150.0) if a.algo in ("lcmv", "gss") else ()
p = make_params(a.algo, n_mics=a.mics, interf=interf, hop=a.hop)
lay = BF_PLANAR if a.layout == "planar" else BF_INTERLEAVED
bf = Beamformer(p, n_streams=a.streams, layout=lay, n_dirs=a.dirs, das_impl=1 if a.das_f64 else 0, precision=1 if a.mixed else 0)
if a.dirs > 1:
    bf.set_thetas([-90.0 + 180.0 * d / (a.dirs - 1) for d in range(a.dirs)])
shape = (a.streams, a.mics, a.frames * a.hop) if lay == BF_PLANAR else (a.streams, a.frames * a.hop, a.mics)
x = torch.rand(shape, device="cuda") - 0.5
y = torch.empty((a.streams * a.dirs, a.frames * a.hop), device="cuda")
import time
t0 = time.perf_counter()
n_warm = 0
while n_warm < a.warmup or (time.perf_counter() - t0) < a.settle_ms * 1e-3:  # clocks settle ~50 launches after idle
    bf.process_device(x.data_ptr(), a.frames, y.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    n_warm += 1
    if n_warm % 8 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
ms, msk = bf.time_device(x.data_ptr(), a.frames, y.data_ptr(), a.iters, torch.cuda.current_stream().cuda_stream)
print(f"{a.algo} M={a.mics} F={a.frames} S={a.streams} D={a.dirs}: call {ms:.4f} ms, kernel {msk:.4f} ms, "
      f"{a.frames*a.streams/ms/1e3:.3f} Mframes/s in, {a.frames*a.streams*a.dirs/ms/1e3:.3f} M beam-frames/s out")
