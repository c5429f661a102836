#!/usr/bin/env python3
"""HBM-side traffic of a kernel chain (stft -> per-bin -> istft, or the one fused kernel) from two rocprofv3 --pmc passes.

Usage: pmc_traffic_chain.py <calib_fetch_dir> <calib_write_dir> <fetch_dir> <write_dir> <step_kernel_substr> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB, per dispatch.  They come from the L2's fabric-side request counters: Infinity-Cache hits
are counted too (MI355X_MICROARCH.md), so this is "bytes that left the XCD's L2", an upper bound of the HBM bytes.
gfx950 reports 1/2 of a wide coalesced read: the factor is calibrated on a known 1 GiB stream (tools/ubench/fetch_calib.hip).
A "step" = one dispatch of the kernel whose name contains <step_kernel_substr> (the chain's first kernel)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time


def per_kernel(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


cf, cw, kf, kw, step, out = sys.argv[1:7]
GiB = float(1 << 30)
calf, calw = per_kernel(cf, "FETCH_SIZE"), per_kernel(cw, "WRITE_SIZE")
mean = lambda v: sum(v) / len(v)
pick = lambda acc, sub: mean([x for k, v in acc.items() if sub in k for x in v])
fetch_factor = GiB / (pick(calf, "copy_x4") * 1024.0)      # 16 B / lane, the bin pipeline's access width
fetch_factor_dword = GiB / (pick(calf, "read_dword") * 1024.0)
write_factor = GiB / (pick(calw, "copy_x4") * 1024.0)
try:  # 12-byte accesses (z48 spectra of the mvdr / lcmv chain): reported beside the 16-byte factors, which the totals use
    x3_bytes = float((1 << 30) // 12 * 12)
    fetch_factor_x3 = x3_bytes / (pick(calf, "copy_x3") * 1024.0)
    write_factor_x3 = x3_bytes / (pick(calw, "copy_x3") * 1024.0)
except Exception:
    fetch_factor_x3 = write_factor_x3 = None
F, W = per_kernel(kf, "FETCH_SIZE"), per_kernel(kw, "WRITE_SIZE")
ours = sorted(k for k in set(F) | set(W) if "bf::" in k or "das_fused" in k)
steps = max(len(v) for k, v in F.items() if step in k)
def git_head():
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


# the git hash: of the tree the numbers were taken on (the GPU box has no .git: BF_GIT_HEAD, exported by the job script, or the
# repo's own HEAD when this runs in the checkout)
res = {"git_head": os.environ.get("BF_GIT_HEAD") or git_head(), "taken_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
       "kernel_names": ours,   # rocprofv3's full spelling; bench.py compares them (normalised) with the kernels it launches
       "calibration": {"known_bytes": GiB, "fetch_factor_x4": fetch_factor, "fetch_factor_dword": fetch_factor_dword,
                       "write_factor_x4": write_factor, "fetch_factor_x3": fetch_factor_x3, "write_factor_x3": write_factor_x3},
       "steps_profiled": steps, "kernels": {}}
tr = tw = 0.0
for k in ours:
    r = sum(F.get(k, [])) * 1024.0 * fetch_factor / steps
    w = sum(W.get(k, [])) * 1024.0 * write_factor / steps
    res["kernels"][k[:90]] = {"dispatches_per_step": len(F.get(k, [])) / steps, "read_bytes_per_step": r, "write_bytes_per_step": w}
    tr += r
    tw += w
res["hbm_read_bytes_per_launch"] = tr
res["hbm_write_bytes_per_launch"] = tw
res["hbm_bytes_per_launch"] = tr + tw
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
