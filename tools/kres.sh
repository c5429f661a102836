#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra flags]   -> per-kernel VGPRs / spills / LDS / occupancy from hipcc's resource remarks
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -DBF_NFFT=1024 "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | \
  grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-260
