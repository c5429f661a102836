cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_hops_gpu.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/time_hops.py 2>&1 | grep -v amdgpu | grep "1024\|256"
