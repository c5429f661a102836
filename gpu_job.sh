cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_controllers_gpu.py -x -q 2>&1 | tail -8
