#!/bin/bash
# The driver-shaped bench line un-profiled and under the kernel trace (steps 1 / 1b of tools/gpu_profile_all.sh), plus the kernel split of the
# nodes at the JACK periods other than 512: tools/gpu_profile_bench.sh <tag>   -> gpurun_out/<tag>_*
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${tag}_bench_das8.json 2> gpurun_out/${tag}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_bench -- python bench.py --no-cpu > gpurun_out/${tag}_bench_das8_profiled.json 2>> gpurun_out/${tag}_bench.err
for f in $(find gpurun_out/${tag}_trace_bench -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_bench_kernel_stats.csv; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace_bench_ne -- python bench.py --no-cpu --no-extra > gpurun_out/${tag}_bench_das8_noextra_profiled.json 2>> gpurun_out/${tag}_bench.err
for f in $(find gpurun_out/${tag}_trace_bench_ne -name "*kernel_stats*"); do cp $f gpurun_out/${tag}_bench_noextra_kernel_stats.csv; done
CFGS="das 256 131072
das 64 524288
das 1024 32768
phase 256 131072
mvdr 256 131072
phase 1024 32768
mvdr 1024 32768" bash tools/prof_hops.sh > gpurun_out/${tag}_periods_kernel_split.txt 2>&1
head -3 gpurun_out/${tag}_bench_noextra_kernel_stats.csv | cut -c1-200
cat gpurun_out/${tag}_periods_kernel_split.txt
