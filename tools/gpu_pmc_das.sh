#!/bin/bash
# usage: tools/gpu_pmc_das.sh <tag>   (the fused fp32 das kernel)
tag=$1
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/${tag}_p$i -- python tools/run_das.py --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_p$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${tag}_p1 gpurun_out/${tag}_p2 gpurun_out/${tag}_p3 gpurun_out/${tag}_p4 gpurun_out/${tag}_p5 gpurun_out/${tag}_p6 gpurun_out/${tag}_p7 gpurun_out/${tag}_p8 | grep das_fused | cut -c62- > gpurun_out/${tag}_pmc.txt
cat gpurun_out/${tag}_pmc.txt
