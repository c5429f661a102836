cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/r03_h_bench_das8.json 2> gpurun_out/r03_h_bench.err; tail -1 gpurun_out/r03_h_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_h_trace_ne -- python bench.py --no-cpu --no-extra > gpurun_out/r03_h_bench_das8_noextra_profiled.json 2>> gpurun_out/r03_h_bench.err
for f in $(find gpurun_out/r03_h_trace_ne -name "*kernel_stats*"); do cp $f gpurun_out/r03_h_bench_noextra_kernel_stats.csv; done
head -2 gpurun_out/r03_h_bench_noextra_kernel_stats.csv | cut -c1-160
