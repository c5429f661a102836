cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dirs_shared_gpu.py  -q -m gpu 2>&1 | grep -v amdgpu | tail -5
