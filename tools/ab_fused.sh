#!/bin/bash
# A/B builds of das_fused.hip: tools/ab_fused.sh <tag> [extra -D flags...]  -> beamform_amd/lib/libbfcore_<tag>.so (same ABI; select with BFCORE_LIB)
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c beamform_amd/csrc/das_fused.hip -o build/ab/das_fused_$tag.o
objs=$(ls build/obj/*.o | grep -v "/das_fused.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o beamform_amd/lib/libbfcore_$tag.so $objs build/ab/das_fused_$tag.o
echo built beamform_amd/lib/libbfcore_$tag.so
