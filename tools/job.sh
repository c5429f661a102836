cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_hops_gpu.py tests/test_shard_gpu.py tests/test_golden_gpu.py tests/test_dirs_gpu.py tests/test_variants_gpu.py tests/test_edges_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/time_scene.py mvdr 8 65536 4 | head -2
python tools/time_node.py lcmv 16 32768
python tools/time_node.py mvdr 16 32768
python tools/time_node.py lcmv 8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j6 -- python tools/run_das.py --algo mvdr --iters 10 > gpurun_out/j6.log 2>&1
for f in $(find gpurun_out/j6 -name "*kernel_stats*"); do cut -c1-150 $f | head -4; done
