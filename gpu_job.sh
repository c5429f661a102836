cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hops_gpu.py -q > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
