cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hops_gpu.py tests/test_das_gpu.py tests/test_node_shim_gpu.py -q > gpurun_out/t1.log 2>&1; tail -12 gpurun_out/t1.log
