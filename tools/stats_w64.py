#!/usr/bin/env python3
"""Overlap-add hand-off statistics of das_f64_pair_kernel from a -DBF_W64_STATS build (tools/ab_w64.sh stats -DBF_W64_STATS; BFCORE_LIB=...)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beamform_amd import capi
from beamform_amd.capi import BF_DAS_F64, Beamformer
from beamform_amd.params import make_params
M, F = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = torch.rand((M, F * 512), device="cuda") - 0.5
y = torch.empty(F * 512, device="cuda")
bf = Beamformer(make_params("das", n_mics=M), das_impl=BF_DAS_F64)
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
lib = ctypes.CDLL(capi.LIB_PATH)
print("time:", bf.time_device(x.data_ptr(), F, y.data_ptr(), 10, s))
st = np.zeros((256, 8, 5), dtype=np.uint64)
assert lib.bf_dbg_stats(st.ctypes.data_as(ctypes.c_void_p)) == 0
st = st.astype(np.float64)
print(f"{F} frames, last launch, per wavefront index (mean over blocks): boundaries where this side came first / second, us waited for the other side's store acknowledgement, us in the kernel")
for w in range(8):
    print(f"  w{w}: first {st[:, w, 0].mean() + st[:, w, 1].mean():5.1f}  second {st[:, w, 2].mean():5.1f}  waited {st[:, w, 3].mean() * 0.01:6.2f} us of {st[:, w, 4].mean() * 0.01:7.2f}")
