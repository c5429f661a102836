cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k other_bands 2>&1 | tail -3
BF_MVDR_GROUP=1 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "other_bands and (mvdr-8 or lcmv-8)" 2>&1 | tail -2
