cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_fused_bins_gpu.py tests/test_golden_gpu.py tests/test_dirs_gpu.py tests/test_hops_gpu.py tests/test_variants_gpu.py tests/test_edges_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/run_das.py --algo phase --iters 10 | tail -1
BF_FUSED_BINS=2 python tools/run_das.py --algo das --das-f64 --iters 10 | tail -1
python tools/run_das.py --algo das --das-f64 --mics 16 --frames 32768 --iters 10 | tail -1
