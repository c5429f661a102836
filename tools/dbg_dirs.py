import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
M, F, D = 4, 5, 8
p = make_params("das", n_mics=M, theta=0.0)
x = make_scene(M, F, seed=1)
thetas = [float(v) for v in np.linspace(-170.0, 175.0, D)]
bf = Beamformer(p, n_dirs=D); bf.set_thetas(thetas)
xd = torch.from_numpy(x).cuda()
yd = torch.empty((D, F * 512), dtype=torch.float32, device="cuda")
bf.process_device(xd.data_ptr(), F, yd.data_ptr()); torch.cuda.synchronize()
y = yd.cpu().numpy()
for d in range(D):
    ref = Beamformer(dict(p, theta=thetas[d])).process(x)
    diff = np.abs(y[d] - ref)
    bad = np.nonzero(diff > 0)[0]
    print(d, float(diff.max()), len(bad), (bad[:5] // 512, bad[:5] % 512) if len(bad) else "", float(np.abs(ref).max()))
