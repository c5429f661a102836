cd /root/repo
export TMPDIR=/tmp
python tools/time_scene.py mvdr 8 65536 4 2>&1 | head -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j5 -- python tools/run_das.py --algo mvdr --iters 10 > gpurun_out/j5.log 2>&1
for f in $(find gpurun_out/j5 -name "*kernel_stats*"); do cut -c1-150 $f | head -4; done
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
