#!/usr/bin/env python3
"""Per-phase wall-clock of the fused DAS kernel (debug library built with -DBF_DAS_STAMPS; see das_fused.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("BFCORE_LIB", os.path.join(ROOT, "beamform_amd", "lib", "libbfcore_stamps.so"))
sys.path.insert(0, ROOT)
import torch
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
M, F = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 65536
p = make_params("das", n_mics=M, theta=20.0)
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand((M, F * 512), device="cuda", generator=g) - 0.5
y = torch.empty(F * 512, device="cuda")
bf = Beamformer(p)
s = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
import ctypes
libc = ctypes.CDLL(None)   # the library reads the switch with getenv(): os.environ alone would do, putenv keeps it explicit
for rep in range(3):
    os.environ["BF_DAS_STAMPS_PRINT"] = "1"
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)   # read out + zero
    torch.cuda.synchronize()
    del os.environ["BF_DAS_STAMPS_PRINT"]
    ms, msk = bf.time_device(x.data_ptr(), F, y.data_ptr(), 200, s)
    print(f"kernel {msk:.4f} ms (stamped build, back-to-back launches)")
os.environ["BF_DAS_STAMPS_PRINT"] = "1"
bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
