#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of das in double for several A/B libraries on ONE box (separate --pmc passes; calibration in the same job):
#   tools/gpu_traffic_ab.sh <tag> base nt1 ...   ("base" = libbfcore.so, otherwise libbfcore_<name>.so; NAME:ENV=VAL adds an env setting)
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
P="timeout 300 rocprofv3 --kernel-trace --output-format csv"
mkdir -p gpurun_out
$P --pmc FETCH_SIZE -d gpurun_out/${tag}_cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/${tag}_cal.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/${tag}_cal_w -- ./tools/ubench/fetch_calib.bin >> gpurun_out/${tag}_cal.log 2>&1
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=${spec#*:}
  lib=$PWD/beamform_amd/lib/libbfcore_$name.so; [ "$name" = base ] && lib=$PWD/beamform_amd/lib/libbfcore.so
  key=$(echo "$spec" | tr ':=,' '___')
  export BFCORE_LIB=$lib
  [ -n "$envs" ] && export $envs
  $P --pmc FETCH_SIZE -d gpurun_out/${tag}_${key}_f -- python tools/run_das.py --algo das --das-f64 --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_${key}.log 2>&1
  $P --pmc WRITE_SIZE -d gpurun_out/${tag}_${key}_w -- python tools/run_das.py --algo das --das-f64 --iters 3 --warmup 2 --settle-ms 0 >> gpurun_out/${tag}_${key}.log 2>&1
  $P --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/${tag}_${key}_h -- python tools/run_das.py --algo das --das-f64 --iters 3 --warmup 2 --settle-ms 0 >> gpurun_out/${tag}_${key}.log 2>&1
  [ -n "$envs" ] && unset ${envs%%=*}
  echo "== $spec: $(python tools/pmc_traffic_chain.py gpurun_out/${tag}_cal_f gpurun_out/${tag}_cal_w gpurun_out/${tag}_${key}_f gpurun_out/${tag}_${key}_w das_f64_pair gpurun_out/traffic_${tag}_${key}.json | grep 'hbm_' | tr -d '\n')"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/${tag}_${key}_h/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "das_f64_pair" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
h, m = (sum(acc[k]) / max(1, len(acc[k])) for k in ("TCC_HIT_sum", "TCC_MISS_sum"))
print(f"   TCC hit {h:.3e} miss {m:.3e} hit rate {h / max(1.0, h + m):.3f}")
PY
done
