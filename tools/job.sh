cd /root/repo
export TMPDIR=/tmp
for m in 1 2 3 4 8; do BF_STFT_RUNS=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j8_$m -- python tools/run_das.py --algo mvdr --iters 10 > gpurun_out/j8.log 2>&1; f=$(find gpurun_out/j8_$m -name "*kernel_stats*"); echo -n "x$m: "; grep stft_kernel $f | sed 's/.*)",//' | cut -d, -f1-3; done
