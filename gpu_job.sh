cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu -x > gpurun_out/full_gpu.log 2>&1; tail -6 gpurun_out/full_gpu.log
