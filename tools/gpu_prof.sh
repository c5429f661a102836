#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <tag> <run_das.py args...>
# kernel-trace stats + four PMC passes of tools/run_das.py with the given arguments -> gpurun_out/<tag>_*
tag=$1; shift
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python tools/run_das.py "$@" > gpurun_out/${tag}_trace.log 2>&1
for f in $(find gpurun_out/${tag}_trace -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_kernel_stats.csv; done
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/${tag}_p$i -- python tools/run_das.py "$@" --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_p$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${tag}_p1 gpurun_out/${tag}_p2 gpurun_out/${tag}_p3 gpurun_out/${tag}_p4 > gpurun_out/${tag}_pmc.txt
cut -c1-150 gpurun_out/${tag}_kernel_stats.csv | head -8
cat gpurun_out/${tag}_pmc.txt | head -60
tail -1 gpurun_out/${tag}_trace.log
