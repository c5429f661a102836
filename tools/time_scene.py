#!/usr/bin/env python3
"""Time a node on uniform noise and on the tiled synthetic scene: tools/time_scene.py <algo> [mics] [frames] [reps]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
algo = sys.argv[1] if len(sys.argv) > 1 else "mvdr"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
F = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
interf = (-60.0, 90.0, 150.0) if algo in ("lcmv", "gss") else ()
p = make_params(algo, n_mics=M, interf=interf)
g = torch.Generator(device="cuda").manual_seed(7)
xn = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
xs = torch.from_numpy(make_scene(M, 2048, seed=77)).cuda().repeat(1, F // 2048).contiguous()
y = torch.empty(F * 512, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, x in (("noise", xn), ("scene", xs), ("noise", xn), ("scene", xs)):
    bf = Beamformer(p)
    for _ in range(3):
        bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
    torch.cuda.synchronize()
    ts = [bf.time_device(x.data_ptr(), F, y.data_ptr(), 5, s)[0] for _ in range(reps)]
    print(f"{algo} {M}-mic {F} frames, {name}: " + " ".join(f"{t:.3f}" for t in ts) + " ms")
    bf.close()
