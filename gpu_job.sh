cd /root/repo
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_fused_bins_gpu.py -x -q -k "das_f64" 2>&1 | tail -2
for i in 1 2; do timeout 120 python tools/time_das_f64.py 2>&1 | tail -1; done
