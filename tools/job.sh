cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_variants_gpu.py tests/test_shard_gpu.py tests/test_hops_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/fuzz_parity.py 51 200 2>&1 | tail -1
