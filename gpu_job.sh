cd /root/repo
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mvdr2 -- python tools/run_das.py --algo mvdr --iters 5 > gpurun_out/prof_mvdr2.log 2>&1
for f in $(find gpurun_out/prof_mvdr2 -name "*kernel_stats*"); do cut -c1-150 $f | head -5; done
