cd /root/repo
python -m pytest tests/test_hops_gpu.py tests/test_shard_gpu.py tests/test_dirs_gpu.py -x -q -m gpu 2>&1 | tail -5
