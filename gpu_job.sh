cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fused_bins_gpu.py tests/test_variants_gpu.py tests/test_golden_gpu.py tests/test_dirs_gpu.py tests/test_edges_gpu.py -q > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
