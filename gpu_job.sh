cd /root/repo
export TMPDIR=/tmp
export BF_DAS_W64=1
P="rocprofv3 --kernel-trace --output-format csv"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/w$i -- python tools/run_das.py --iters 3 > gpurun_out/w$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/w1 gpurun_out/w2 gpurun_out/w3 | cut -c62-
