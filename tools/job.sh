cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_das_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do python tools/run_das.py --algo das --layout interleaved --iters 30 | tail -1; done
