cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r03_d
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_bench_ne -- python bench.py --no-cpu --no-extra > gpurun_out/${tag}_bench_das8_noextra_profiled.json 2>> gpurun_out/${tag}_bench.err
for f in $(find gpurun_out/${tag}_trace_bench_ne -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_bench_noextra_kernel_stats.csv; done
head -3 gpurun_out/${tag}_bench_noextra_kernel_stats.csv
cat gpurun_out/${tag}_bench_das8_noextra_profiled.json | cut -c1-600
