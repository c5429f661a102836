cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_dirs_shared_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_d_driver_shape.json 2> gpurun_out/r03_d_driver_shape.err
timeout 900 bash tools/gpu_profile_all.sh r03_d 2>&1 | grep -v simple_timer | grep "hbm_bytes_per_launch"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_d_dirs16_trace -- python tools/run_das.py --algo das --dirs 16 --iters 10 > gpurun_out/r03_d_dirs16.log 2>&1
for f in $(find gpurun_out/r03_d_dirs16_trace -name "*kernel_stats*"); do cp $f gpurun_out/r03_d_das8_dirs16_kernel_stats.csv; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_d_lcmv8_trace -- python tools/run_das.py --algo lcmv --mics 8 --iters 10 > gpurun_out/r03_d_lcmv8.log 2>&1
for f in $(find gpurun_out/r03_d_lcmv8_trace -name "*kernel_stats*"); do cp $f gpurun_out/r03_d_lcmv8_kernel_stats.csv; done
tail -1 gpurun_out/r03_d_dirs16.log gpurun_out/r03_d_lcmv8.log
