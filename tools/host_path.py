import sys, time, numpy as np
sys.path.insert(0, '.')
from beamform_amd.capi import Beamformer, host_array
from beamform_amd.params import make_params
F = 65536
p = make_params("das", n_mics=8)
bf = Beamformer(p)
x = (np.random.default_rng(0).random((8, F * 512), dtype=np.float32) - 0.5)
for i in range(3):
    t0 = time.perf_counter(); y = bf.process(x); dt = time.perf_counter() - t0
    print(f"host path: {dt*1e3:.1f} ms per {F} frames = {F/dt/1e6:.2f} Mframes/s, {x.nbytes/dt/1e9:.1f} GB/s in")
xp = host_array(x.shape); xp[...] = x
yp = host_array((F * 512,))
for i in range(3):
    t0 = time.perf_counter(); y2 = bf.process(xp, out=yp); dt = time.perf_counter() - t0
    print(f"host path, page-locked buffers: {dt*1e3:.1f} ms per {F} frames = {F/dt/1e6:.2f} Mframes/s, {x.nbytes/dt/1e9:.1f} GB/s in")
