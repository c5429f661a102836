cd /root/repo
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/q$i -- python tools/run_das.py --iters 3 > gpurun_out/q$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/q1 gpurun_out/q2 gpurun_out/q3 gpurun_out/q4 gpurun_out/q5 gpurun_out/q6
grep kernel gpurun_out/q1.log | tail -1
