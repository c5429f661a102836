cd /root/repo
export TMPDIR=/tmp
for m in 1 2 3 4 6 8; do echo "slots x$m"; BF_ISTFT_SLOTS=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j7_$m -- python tools/run_das.py --algo mvdr --iters 10 > gpurun_out/j7.log 2>&1; for f in $(find gpurun_out/j7_$m -name "*kernel_stats*"); do grep istft32 $f | cut -d, -f1-4 | cut -c60-; done; done
