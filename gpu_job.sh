cd /root/repo
export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d gpurun_out/cal_f -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_f.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/cal_w -- ./tools/ubench/fetch_calib.bin > gpurun_out/cal_w.log 2>&1
$P --pmc FETCH_SIZE -d gpurun_out/das_f2 -- python tools/run_das.py --iters 5 > gpurun_out/das_f2.log 2>&1
$P --pmc WRITE_SIZE -d gpurun_out/das_w2 -- python tools/run_das.py --iters 5 > gpurun_out/das_w2.log 2>&1
python tools/pmc_traffic.py gpurun_out/cal_f gpurun_out/cal_w gpurun_out/das_f2 gpurun_out/das_w2 das_fused_kernel gpurun_out/traffic_das8.json | tail -7
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c -- python bench.py --steps 30 --warmup 5 > gpurun_out/bench_r01_c.json 2> gpurun_out/bench_r01_c.err
for f in $(find gpurun_out/prof_c -name "*kernel_stats*"); do cut -c1-160 $f | head -3; done
cut -c1-1400 gpurun_out/bench_r01_c.json
