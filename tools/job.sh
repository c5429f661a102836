cd /root/repo
python -m pytest tests/test_dirs_shared_gpu.py -x -q -m gpu 2>&1 | tail -2
