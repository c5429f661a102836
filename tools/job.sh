cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_f_driver_shape.json 2> gpurun_out/r03_f_driver_shape.err
timeout 900 bash tools/gpu_profile_all.sh r03_f 2>&1 | grep "hbm_bytes_per_launch"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_f_dirs16_trace -- python tools/run_das.py --algo das --dirs 16 --iters 10 > gpurun_out/r03_f_dirs16.log 2>&1
for f in $(find gpurun_out/r03_f_dirs16_trace -name "*kernel_stats*"); do cp $f gpurun_out/r03_f_das8_dirs16_kernel_stats.csv; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_f_lcmv8_trace -- python tools/run_das.py --algo lcmv --mics 8 --iters 10 > gpurun_out/r03_f_lcmv8.log 2>&1
for f in $(find gpurun_out/r03_f_lcmv8_trace -name "*kernel_stats*"); do cp $f gpurun_out/r03_f_lcmv8_kernel_stats.csv; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_f_il8_trace -- python tools/run_das.py --algo das --layout interleaved --iters 50 > gpurun_out/r03_f_il8.log 2>&1
for f in $(find gpurun_out/r03_f_il8_trace -name "*kernel_stats*"); do cp $f gpurun_out/r03_f_das8_il_kernel_stats.csv; done
echo done
