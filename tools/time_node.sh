#!/bin/bash
# time one node's chain for several A/B libraries on ONE box: tools/time_node.sh "<run_das.py args>" tagA tagB ...   ("base" = libbfcore.so)
cd "$(dirname "$0")/.."
args=$1; shift
for rep in 1 2 3; do
  for tag in "$@"; do
    lib=beamform_amd/lib/libbfcore_$tag.so; [ "$tag" = base ] && lib=beamform_amd/lib/libbfcore.so
    BFCORE_LIB=$PWD/$lib timeout 300 python tools/run_das.py $args 2>&1 | tail -1 | sed "s/^/$tag: /" | cut -c1-140
  done
done
