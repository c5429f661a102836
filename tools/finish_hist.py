#!/usr/bin/env python3
"""When the blocks / wavefronts of das_f64_pair_kernel finish (a -DBF_W64_STAMPS build: tools/ab_w64.sh stamps -DBF_W64_STAMPS;
BFCORE_LIB=.../libbfcore_stamps.so).  s_memrealtime (100 MHz) at kernel entry and exit per wavefront, frame pairs taken, XCC id.
tools/finish_hist.py [frames] [out.json]"""
import ctypes, json, os, sys
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
for _ in range(60):
    bf.process_device(x.data_ptr(), F, y.data_ptr(), 0, s)
torch.cuda.synchronize()
lib = ctypes.CDLL(capi.LIB_PATH)
st = np.zeros((256, 8, 64), dtype=np.uint64)
assert lib.bf_dbg_stamps(st.ctypes.data_as(ctypes.c_void_p)) == 0
used = st[:, 0, 63] > 0   # blocks of this launch's grid
st = st[used]
t0 = int(st[:, :, 0].min())
entry = (st[:, :, 0].astype(np.int64) - t0) * 0.01
end = (st[:, :, 63].astype(np.int64) - t0) * 0.01          # us
pairs = st[:, :, 62].astype(np.int64)
xcc = st[:, 0, 61].astype(np.int64) & 15
bend = end.max(axis=1)
first_wave = end.min(axis=1)
print(f"{F} frames: kernel span {bend.max():.1f} us; block end min {bend.min():.1f} mean {bend.mean():.1f} max {bend.max():.1f} "
      f"(spread {100 * (bend.max() - bend.min()) / bend.max():.1f} % of the span; mean is {100 * (1 - bend.mean() / bend.max()):.1f} % under max)")
print(f"inside a block: first wavefront done {np.mean(bend - first_wave):.1f} us before the last (mean), max {np.max(bend - first_wave):.1f}; "
      f"mean wavefront idle at the block's end {np.mean(bend[:, None] - end):.1f} us")
print(f"SIMD idle before kernel end (mean over wavefronts of kernel_end - wave_end): {np.mean(bend.max() - end):.1f} us = {100 * np.mean(bend.max() - end) / bend.max():.1f} %")
print("pairs per wavefront slot (mean):", [round(float(pairs[:, w].mean()), 1) for w in range(8)], "pairs per block: min", pairs.sum(axis=1).min(), "max", pairs.sum(axis=1).max())
print("per XCC: blocks, mean block end, mean pairs per block")
for c in range(8):
    m = xcc == c
    if m.any():
        print(f"  xcc {c}: {int(m.sum()):3d} blocks, end {bend[m].mean():7.1f} us (min {bend[m].min():.1f} max {bend[m].max():.1f}), pairs {pairs.sum(axis=1)[m].mean():.1f}")
clk = (st[:, :, 60].astype(np.int64) - st[:, :, 59].astype(np.int64)) / np.maximum(1, st[:, :, 63].astype(np.int64) - st[:, :, 0].astype(np.int64)) * 0.1  # GHz
nb = len(st)
print(f"shader clock inside the kernel (s_memtime / s_memrealtime): mean {clk[:nb].mean():.3f} GHz; per XCC:", [round(float(clk[:nb][xcc[:nb] == c].mean()), 3) if (xcc[:nb] == c).any() else None for c in range(8)])
hist, edges = np.histogram(bend, bins=12)
print("block-end histogram (us):", [f"{edges[i]:.0f}-{edges[i + 1]:.0f}: {int(hist[i])}" for i in range(len(hist))])
if len(sys.argv) > 2:
    json.dump({"frames": F, "span_us": float(bend.max()), "block_end_us": [float(v) for v in bend], "xcc": [int(v) for v in xcc],
               "pairs_per_block": [int(v) for v in pairs.sum(axis=1)], "wave_end_us": end.tolist(), "entry_us": entry.tolist()}, open(sys.argv[2], "w"))
