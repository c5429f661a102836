cd /root/repo
export TMPDIR=/tmp
tag=r03_c
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d gpurun_out/${tag}_cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/${tag}_cal.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/${tag}_cal_w -- ./tools/ubench/fetch_calib.bin >> gpurun_out/${tag}_cal.log 2>&1
name=das8_f64; step=das_f64_fused
$P --pmc FETCH_SIZE -d gpurun_out/${tag}_${name}_f -- python tools/run_das.py --algo das --das-f64 --iters 3 --warmup 2 --settle-ms 0 > gpurun_out/${tag}_${name}.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/${tag}_${name}_w -- python tools/run_das.py --algo das --das-f64 --iters 3 --warmup 2 --settle-ms 0 >> gpurun_out/${tag}_${name}.log 2>&1
python tools/pmc_traffic_chain.py gpurun_out/${tag}_cal_f gpurun_out/${tag}_cal_w gpurun_out/${tag}_${name}_f gpurun_out/${tag}_${name}_w $step gpurun_out/traffic_${name}.json | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${name}_trace -- python tools/run_das.py --algo das --das-f64 --iters 10 >> gpurun_out/${tag}_${name}.log 2>&1
for f in $(find gpurun_out/${tag}_${name}_trace -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_${name}_kernel_stats.csv; done
cut -c1-140 gpurun_out/${tag}_${name}_kernel_stats.csv | head -4
bash tools/gpu_prof.sh r03_das8_f64 --algo das --das-f64 > gpurun_out/r03_das8_f64_prof.txt 2>&1
grep das_f64 gpurun_out/r03_das8_f64_pmc.txt | cut -c62-
