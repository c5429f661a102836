cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hops_gpu.py -x -q 2>&1 | tail -4
