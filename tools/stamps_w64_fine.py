#!/usr/bin/env python3
"""Phases inside step 8 of das_f64_pair_kernel (tools/ab_w64.sh fine -DBF_W64_STAMPS -DBF_W64_FINE; BFCORE_LIB=.../libbfcore_fine.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd import capi
from beamform_amd.capi import BF_DAS_BINS_F64, Beamformer
from beamform_amd.params import make_params
M, F = 8, 65536
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
t = st.astype(np.int64) * 0.01  # us
names = ["window fold + next loads issued", "tw1 + first 16-pt + tw mult (+flag wait at m=0)", "T1 exchange", "second 16-pt + tw2", "T2 swaps", "gains + 4-pt + S += (to next mic)"]
for m in range(M):
    d = [t[:, :, 6 * m + k + 1] - t[:, :, 6 * m + k] for k in range(5)]
    nxt = t[:, :, 6 * (m + 1)] if m + 1 < M else t[:, :, 48]
    d.append(nxt - t[:, :, 6 * m + 5])
    print(f"mic {m}: " + "  ".join(f"{x.mean():5.2f}" for x in d) + f"   total {(nxt - t[:, :, 6 * m]).mean():5.2f} us")
print("columns:", "; ".join(names))
b = [t[:, :, 48 + k + 1] - t[:, :, 48 + k] for k in range(5)]
print("backward: tw2 load + 4-pt + T2 %.2f; mult + 16-pt %.2f; T1 %.2f; mult + 16-pt %.2f; epilogue (hand-off, stores) %.2f" % tuple(x.mean() for x in b))
