#!/usr/bin/env python3
"""Race hunt for das_f64_ring_kernel ([sample][mic] input through the blocks' hop rings): the same batch many times against the planar
kernel's output, which must be matched bit for bit (a stale or half-written ring slot shows up as a different float).
tools/stress_das_f64_ring.py [reps] [frames] [mics] [streams]; BF_DAS_F64_SCHED="5,3,1" makes every second pair open a chunk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from beamform_amd.capi import BF_DAS_F64, BF_INTERLEAVED, Beamformer
from beamform_amd.params import make_params
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
F = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
p = make_params("das", n_mics=M, theta=35.0)
g = torch.Generator(device="cuda").manual_seed(21)
x = torch.rand(S, M, F * 512, device="cuda", generator=g) - 0.5
xi = x.transpose(1, 2).contiguous()
y0 = torch.empty(S, F * 512, device="cuda")
y = torch.empty(S, F * 512, device="cuda")
Beamformer(p, n_streams=S, das_impl=BF_DAS_F64).process_device(x.data_ptr(), F, y0.data_ptr())
torch.cuda.synchronize()
n_bad = 0
for rep in range(reps):
    bf = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
    y.fill_(float("nan"))
    bf.process_device(xi.data_ptr(), F, y.data_ptr())
    torch.cuda.synchronize()
    bad = torch.nonzero(~((y == y0).view(S * F, 512).all(dim=1))).flatten().cpu().numpy()
    if len(bad):
        n_bad += 1
        print(f"rep {rep}: {len(bad)} hops differ, first {bad[:16]}", flush=True)
    bf.close()
print(f"ring kernel, {reps} repetitions of {S} x {F} frames of {M} microphones: {n_bad} with differing hops")
