cd /root/repo
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d gpurun_out/cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_f.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/cal_w -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_w.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/das_f3 -- python tools/run_das.py --iters 5 > gpurun_out/das_f3.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/das_w3 -- python tools/run_das.py --iters 5 > gpurun_out/das_w3.log 2>&1
python tools/pmc_traffic.py gpurun_out/cal_f gpurun_out/cal_w gpurun_out/das_f3 gpurun_out/das_w3 das_fused_kernel gpurun_out/traffic_das8.json | tail -4
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  $P --pmc $set -d gpurun_out/f$i -- python tools/run_das.py --iters 3 --warmup 2 > gpurun_out/f$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/f1 gpurun_out/f2 gpurun_out/f3 gpurun_out/f4 | cut -c62- > gpurun_out/pmc_final.txt; cat gpurun_out/pmc_final.txt | head -30
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f -- python bench.py > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err
for f in $(find gpurun_out/prof_f -name "*kernel_stats*"); do cut -c1-160 $f | head -4; done
python -c "
import json; d=json.load(open('gpurun_out/bench_f.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['extra']['mvdr_frames_per_s'])"
