#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 --pmc passes, calibrated as MI355X_MICROARCH.md asks.

Usage: pmc_traffic.py <calib_fetch_dir> <calib_write_dir> <kernel_fetch_dir> <kernel_write_dir> <kernel_substr> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB.  The calibration run (tools/ubench/fetch_calib.hip) moves a known 1 GiB with
the same access width as the kernel (one dword per lane, 128-B lines), which gives the factor to apply."""
import csv, glob, json, sys

def mean_counter(d, name, kernel_substr):
    vals = []
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None

cf, cw, kf, kw, ksub, out = sys.argv[1:7]
GiB = float(1 << 30)
rd = mean_counter(cf, "FETCH_SIZE", "read_dword")
cpf = mean_counter(cf, "FETCH_SIZE", "copy_dword")
cpw = mean_counter(cw, "WRITE_SIZE", "copy_dword")
x4f = mean_counter(cf, "FETCH_SIZE", "copy_x4")
x4w = mean_counter(cw, "WRITE_SIZE", "copy_x4")
fetch_factor = GiB / (rd * 1024.0)          # true bytes per reported byte, dword-per-lane reads
write_factor = GiB / (cpw * 1024.0)
kfetch = mean_counter(kf, "FETCH_SIZE", ksub)
kwrite = mean_counter(kw, "WRITE_SIZE", ksub)
res = {
    "kernel": ksub,
    "calibration": {"known_bytes": GiB, "FETCH_SIZE_KiB_read_dword": rd, "FETCH_SIZE_KiB_copy_dword": cpf,
                    "WRITE_SIZE_KiB_copy_dword": cpw, "FETCH_SIZE_KiB_copy_x4": x4f, "WRITE_SIZE_KiB_copy_x4": x4w,
                    "fetch_factor_dword": fetch_factor, "write_factor_dword": write_factor},
    "FETCH_SIZE_KiB": kfetch, "WRITE_SIZE_KiB": kwrite,
    "hbm_read_bytes_per_launch": kfetch * 1024.0 * fetch_factor,
    "hbm_write_bytes_per_launch": kwrite * 1024.0 * write_factor,
}
res["hbm_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
