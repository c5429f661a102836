#!/bin/bash
# Round profile set, run on the GPU box from the repo root: tools/gpu_profile_all.sh <tag>   -> gpurun_out/<tag>_*
# ONLY="mvdr8 lcmv16" tools/gpu_profile_all.sh <tag>: the calibration and the traffic / kernel stats of these chains only
tag=$1
export TMPDIR=/tmp
# the tree the numbers belong to (the GPU box has no .git: gpu_job.sh exports BF_GIT_HEAD before calling this)
export BF_GIT_HEAD=${BF_GIT_HEAD:-$(git rev-parse --short=12 HEAD 2>/dev/null)}
P="timeout 300 rocprofv3 --kernel-trace --output-format csv"
mkdir -p gpurun_out
# 1. the driver-shaped bench line, un-profiled and under the kernel trace
if [ -z "$ONLY" ]; then
timeout 900 python bench.py > gpurun_out/${tag}_bench_das8.json 2> gpurun_out/${tag}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_bench -- python bench.py --no-cpu > gpurun_out/${tag}_bench_das8_profiled.json 2>> gpurun_out/${tag}_bench.err
for f in $(find gpurun_out/${tag}_trace_bench -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_bench_kernel_stats.csv; done
# 1b. the same without the secondary lines: every das_fused_kernel launch in this trace is a full 65 536-frame batch (the streaming-callback
#     line above launches the same kernel on single hops, which drags its average down)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_bench_ne -- python bench.py --no-cpu --no-extra > gpurun_out/${tag}_bench_das8_noextra_profiled.json 2>> gpurun_out/${tag}_bench.err
for f in $(find gpurun_out/${tag}_trace_bench_ne -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_bench_noextra_kernel_stats.csv; done
fi
# 2. calibration of FETCH_SIZE / WRITE_SIZE
$P --pmc FETCH_SIZE -d gpurun_out/${tag}_cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/${tag}_cal.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/${tag}_cal_w -- ./tools/ubench/fetch_calib.bin >> gpurun_out/${tag}_cal.log 2>&1
# 3. traffic of every BASELINE config's chain
run() {  # name, step kernel, run_das args...
  name=$1; step=$2; shift 2
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  $P --pmc FETCH_SIZE -d gpurun_out/${tag}_${name}_f -- python tools/run_das.py "$@" --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_${name}.log 2>&1
  $P --pmc WRITE_SIZE -d gpurun_out/${tag}_${name}_w -- python tools/run_das.py "$@" --iters 3 --warmup 2 --settle-ms 0 >> gpurun_out/${tag}_${name}.log 2>&1
  python tools/pmc_traffic_chain.py gpurun_out/${tag}_cal_f gpurun_out/${tag}_cal_w gpurun_out/${tag}_${name}_f gpurun_out/${tag}_${name}_w $step gpurun_out/traffic_${name}.json | tail -4
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name}_trace -- python tools/run_das.py "$@" --iters 10 >> gpurun_out/${tag}_${name}.log 2>&1
  for f in $(find gpurun_out/${tag}_${name}_trace -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_${name}_kernel_stats.csv; done
  tail -1 gpurun_out/${tag}_${name}.log
}
run das8_f64 das_f64_pair --algo das --das-f64
run das8 das_fused --algo das
run mvdr8 stft_kernel --algo mvdr
run phase8 stft_bins_w64 --algo phase
run phasempf8 stft_bins_w64 --algo phasempf --streams 256 --frames 256
run lcmv16 stft_kernel --algo lcmv --mics 16 --frames 32768
run mvdr8_mixed stft_kernel --algo mvdr --mixed
run lcmv16_mixed stft_kernel --algo lcmv --mics 16 --frames 32768 --mixed
# 4. SQ / LDS counters of the headline kernel (das in double), separate passes
if [ -n "$ONLY" ]; then exit 0; fi
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
