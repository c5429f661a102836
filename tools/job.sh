cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_fused_bins_gpu.py tests/test_edges_gpu.py tests/test_threads_gpu.py tests/test_node_shim_gpu.py -x -q -m gpu 2>&1 | tail -6
