cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_variants_gpu.py -x -q -k "Z48 or z48 or W64" 2>&1 | tail -12
