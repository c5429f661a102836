cd /root/repo
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k other_bands 2>&1 | tail -12
