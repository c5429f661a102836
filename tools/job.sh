cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_hops_gpu.py tests/test_shard_gpu.py tests/test_golden_gpu.py tests/test_dirs_gpu.py tests/test_edges_gpu.py -x -q -m gpu 2>&1 | tail -15
python tools/time_node.py mvdr 8
python tools/time_node.py mvdr 6
python tools/time_node.py mvdr 4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/j1_mvdr -- python tools/run_das.py --algo mvdr --iters 10 > gpurun_out/j1_mvdr.log 2>&1
for f in $(find gpurun_out/j1_mvdr -name "*kernel_stats*"); do cut -c1-150 $f | head -5; done
