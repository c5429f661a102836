cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r4h
P="timeout 200 rocprofv3 --kernel-trace --output-format csv"
$P --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU -d gpurun_out/r4h/p1 -- python tools/run_das.py --hop 1024 --frames 32768 --iters 3 --warmup 2 > gpurun_out/r4h/p1.log 2>&1
python tools/pmc_summary.py gpurun_out/r4h/p1 | cut -c40-
