cd /root/repo
export TMPDIR=/tmp
tag=r04_b
P="timeout 300 rocprofv3 --kernel-trace --output-format csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/${tag}_f64pmc$i -- python tools/run_das.py --das-f64 --iters 3 --warmup 2 > gpurun_out/${tag}_f64pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/${tag}_f64pmc1 gpurun_out/${tag}_f64pmc2 gpurun_out/${tag}_f64pmc3 gpurun_out/${tag}_f64pmc4 | cut -c40- > gpurun_out/${tag}_das8_f64_pair_pmc.txt
cat gpurun_out/${tag}_das8_f64_pair_pmc.txt
