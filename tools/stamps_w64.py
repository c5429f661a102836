#!/usr/bin/env python3
"""Timeline of das_f64_pair_kernel from a -DBF_W64_STAMPS build (tools/ab_w64.sh stamps -DBF_W64_STAMPS; BFCORE_LIB=.../libbfcore_stamps.so):
s_memrealtime (100 MHz) per wavefront at run entry, after the table copy, and per step before the backward transform / before the
epilogue / at the end.  tools/stamps_w64.py [frames]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd import capi
from beamform_amd.capi import BF_DAS_BINS_F64, Beamformer
from beamform_amd.params import make_params
M, F = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = torch.rand((M, F * 512), device="cuda") - 0.5
y = torch.empty(F * 512, device="cuda")
bf = Beamformer(make_params("das", n_mics=M), das_impl=BF_DAS_BINS_F64)
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
lib = ctypes.CDLL(capi.LIB_PATH)
st = np.zeros((256, 8, 64), dtype=np.uint64)
assert lib.bf_dbg_stamps(st.ctypes.data_as(ctypes.c_void_p)) == 0
t = (st.astype(np.int64) - int(st[:, :, 0].min())) * 0.01  # us
n_it = (F // 256 + 15) // 16
print(f"frames {F}: {n_it} steps per block")
print(f"entry      : mean {t[:, :, 0].mean():7.2f} max {t[:, :, 0].max():7.2f} us")
print(f"tables done: mean {t[:, :, 1].mean():7.2f} max {t[:, :, 1].max():7.2f} us")
prev = t[:, :, 1]
for it in range(n_it):
    a, b, c = t[:, :, 2 + 3 * it], t[:, :, 3 + 3 * it], t[:, :, 4 + 3 * it]
    print(f"step {it:2d}: forward {(a - prev).mean():6.2f}  backward {(b - a).mean():5.2f}  epilogue {(c - b).mean():5.2f}  step {(c - prev).mean():6.2f} "
          f"(min {(c - prev).min():6.2f} max {(c - prev).max():6.2f})  ends at mean {c.mean():7.2f} max {c.max():7.2f}")
    prev = c
end = t[:, :, 4 + 3 * (n_it - 1)].max(axis=1)
print(f"block end: min {end.min():.2f} mean {end.mean():.2f} max {end.max():.2f} us; per XCD mean:", [round(float(end[i::8].mean()), 1) for i in range(8)])
