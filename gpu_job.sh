cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
python -m pytest tests/test_fused_bins_gpu.py -x -q -k "das_f64" 2>&1 | tail -15
BF_DAS_F64_W64=0 python tools/time_das_f64.py 2>&1 | tail -2
BF_DAS_F64_W64=1 python tools/time_das_f64.py 2>&1 | tail -2
