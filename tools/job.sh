cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_shard_gpu.py tests/test_hops_gpu.py tests/test_variants_gpu.py tests/test_dirs_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/time_node.py lcmv 16 32768
python tools/time_node.py mvdr 16 32768
python tools/time_node.py lcmv 12 32768
