cd /root/repo
export TMPDIR=/tmp
for A in "--algo gss --mics 8 --streams 256 --frames 256" "--algo mcra --mics 1 --streams 256 --frames 256" "--algo gsc --mics 8 --streams 256 --frames 64" "--algo phasempf --mics 8 --streams 256 --frames 256"; do
timeout 300 python tools/run_das.py $A --iters 5 --warmup 2 | tail -1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gss_trace -- python tools/run_das.py --algo gss --mics 8 --streams 256 --frames 256 --iters 5 --warmup 2 > /dev/null 2>&1
for f in $(find gpurun_out/gss_trace -name "*kernel_stats*"); do head -5 $f | cut -c1-160; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gsc_trace -- python tools/run_das.py --algo gsc --mics 8 --streams 256 --frames 64 --iters 5 --warmup 2 > /dev/null 2>&1
for f in $(find gpurun_out/gsc_trace -name "*kernel_stats*"); do head -5 $f | cut -c1-160; done
