import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from beamform_amd.capi import Beamformer, BF_DAS_F64, BF_INTERLEAVED, launch_trace
from beamform_amd.params import make_params
from beamform_amd.synth import make_scene
ok = True
for (M, F, S) in [(8, 33, 1), (4, 27, 1), (2, 40, 1), (8, 1, 1), (8, 2, 1), (8, 700, 1), (4, 64, 3), (8, 4099, 2), (8, 65536, 1)]:
    p = make_params("das", n_mics=M, theta=-50.0)
    xs = np.stack([make_scene(M, F, seed=1200 + 7 * M + s) for s in range(S)]) if F < 60000 else (np.random.default_rng(1).random((S, M, F * 512), dtype=np.float32) - 0.5)
    y = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64).process(xs if S > 1 else xs[0]).reshape(S, -1)
    xi = np.ascontiguousarray(xs.transpose(0, 2, 1))
    bil = Beamformer(p, n_streams=S, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
    with launch_trace() as tr:
        yi = bil.process(xi if S > 1 else xi[0]).reshape(S, -1)
    same = np.array_equal(yi, y)
    print(M, F, S, "same" if same else "DIFF max %g at %s" % (np.abs(yi - y).max(), np.argwhere(yi != y)[:3].tolist()), [k[:40] for k in tr.kernels][:3], flush=True)
    ok &= same
    if S == 1 and F >= 9 and F < 5000:
        bi = Beamformer(p, das_impl=BF_DAS_F64, layout=BF_INTERLEAVED)
        cuts = sorted({0, 2, 2 * (F // 4), F})
        parts = [bi.process(np.ascontiguousarray(xi[0][a * 512:b * 512])) for a, b in zip(cuts[:-1], cuts[1:])]
        same2 = np.array_equal(np.concatenate(parts), yi[0])
        print("   cuts", "same" if same2 else "DIFF", flush=True)
        ok &= same2
print("ALL OK" if ok else "FAILED")
