#!/bin/bash
# A/B builds of one bin-pipeline source at N = 1024: tools/ab_bins.sh <source without .hip> <tag> [extra -D flags...]
#   -> beamform_amd/lib/libbfcore_<tag>.so (same ABI; select with BFCORE_LIB)
set -e
cd "$(dirname "$0")/.."
src=$1; tag=$2; shift 2
mkdir -p build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function -DBF_NFFT=1024 "$@" -c beamform_amd/csrc/$src.hip -o build/ab/${src}_n1024_$tag.o
objs=$(ls build/obj/*.o | grep -v "/${src}_n1024.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o beamform_amd/lib/libbfcore_$tag.so $objs build/ab/${src}_n1024_$tag.o
echo built beamform_amd/lib/libbfcore_$tag.so
