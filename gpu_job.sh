cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_bins_gpu.py tests/test_variants_gpu.py -q -k "one_launch or every_one_launch" > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
timeout 120 python tools/run_das.py --algo das --das-f64 --layout interleaved --iters 20 > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
BF_DAS_F64_PAIR=0 timeout 120 python tools/run_das.py --algo das --das-f64 --layout interleaved --iters 20 > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
timeout 120 python tools/run_das.py --algo das --das-f64 --iters 20 > gpurun_out/t.log 2>&1; tail -1 gpurun_out/t.log
