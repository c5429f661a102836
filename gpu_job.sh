cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_resample_gpu.py tests/test_node_shim_gpu.py -x -q 2>&1 | tail -8
