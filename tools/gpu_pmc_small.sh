#!/bin/bash
# (every pass under its own timeout: a counter set the hardware cannot collect aborts rocprofv3 and leaves the child hanging)
# usage: tools/gpu_pmc_small.sh <tag> <hop>   memory-side counters of the das kernel at one JACK period (the headline batch's samples)
tag=$1; hop=$2
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 90 $P --pmc $set -d gpurun_out/${tag}_p$i -- python tools/run_das.py --hop $hop --frames $((65536*512/hop)) --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_p$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${tag}_p1 gpurun_out/${tag}_p2 gpurun_out/${tag}_p3 gpurun_out/${tag}_p4 gpurun_out/${tag}_p5 gpurun_out/${tag}_p6 gpurun_out/${tag}_p7 gpurun_out/${tag}_p8 | grep das_fused > gpurun_out/${tag}_pmc.txt
cat gpurun_out/${tag}_pmc.txt
