cd /root/repo
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d gpurun_out/cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_f.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/cal_w -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_w.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/das_f -- python tools/run_das.py --iters 5 > gpurun_out/das_f.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/das_w -- python tools/run_das.py --iters 5 > gpurun_out/das_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/cal_f gpurun_out/cal_w gpurun_out/das_f gpurun_out/das_w das_fused_kernel gpurun_out/traffic_das8.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b -- python bench.py --steps 30 --warmup 5 > gpurun_out/bench_r01_b.json 2> gpurun_out/bench_r01_b.err
for f in $(find gpurun_out/prof_b -name "*kernel_stats*"); do cut -c1-160 $f | head -4; done
cut -c1-900 gpurun_out/bench_r01_b.json
