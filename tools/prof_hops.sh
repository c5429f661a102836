#!/bin/bash
# kernel split of the fp64 nodes at the JACK periods other than 512 (the headline batch's samples): rocprofv3 --kernel-trace --stats per configuration
export TMPDIR=/tmp
while read -r algo hop frames; do
  [ -z "$algo" ] && continue
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph_${algo}_$hop -- python tools/run_das.py --algo $algo --hop $hop --frames $frames --iters 5 --warmup 3 --settle-ms 0 > gpurun_out/ph_${algo}_$hop.log 2>&1
  echo "== $algo hop $hop"; for f in $(find gpurun_out/ph_${algo}_$hop -name "*kernel_stats*"); do head -6 $f | cut -d, -f1-4 | cut -c1-170; done
done <<< "${CFGS:-mvdr 256 131072
phase 256 131072
phase 1024 32768
mvdr 1024 32768}"
