#!/bin/bash
# time A/B libraries on ONE box, interleaved: tools/ab_time.sh tagA tagB ... (libbfcore_<tag>.so; "base" = libbfcore.so)
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for tag in "$@"; do
    lib=beamform_amd/lib/libbfcore_$tag.so; [ "$tag" = base ] && lib=beamform_amd/lib/libbfcore.so
    BFCORE_LIB=$PWD/$lib timeout 120 python tools/time_das_f64.py 2>&1 | tail -1 | sed "s/^/$tag: /" | cut -c1-110
  done
done
