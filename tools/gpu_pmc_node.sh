#!/bin/bash
# SQ counters of one node's chain (separate passes, kernel trace only): tools/gpu_pmc_node.sh <tag> "<run_das.py args>"  -> gpurun_out/<tag>_pmc.txt
tag=$1; args=$2
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 $P --pmc $set -d gpurun_out/${tag}_p$i -- python tools/run_das.py $args --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_p$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${tag}_p1 gpurun_out/${tag}_p2 gpurun_out/${tag}_p3 | cut -c30- > gpurun_out/${tag}_pmc.txt
cat gpurun_out/${tag}_pmc.txt
