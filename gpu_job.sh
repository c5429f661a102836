cd /root/repo
export TMPDIR=/tmp
for i in 1 2; do
for alg in "mvdr" ; do
  timeout 200 python tools/run_das.py --algo $alg --iters 10 > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
  BF_STFT_W64=0 timeout 200 python tools/run_das.py --algo $alg --iters 10 > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
done; done
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_shard_gpu.py tests/test_golden_gpu.py tests/test_hops_gpu.py -q > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
