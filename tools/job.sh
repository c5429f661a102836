cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_das_gpu.py tests/test_edges_gpu.py tests/test_shard_gpu.py tests/test_variants_gpu.py -x -q -m gpu 2>&1 | tail -5
python tools/run_das.py --algo das --layout interleaved --iters 20 | tail -1
python tools/run_das.py --algo das --layout interleaved --mics 4 --iters 20 | tail -1
