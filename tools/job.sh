cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo das --das-f64 --iters 50 | tail -1
echo -n "new  "; python tools/run_das.py --algo das --das-f64 --iters 50 | tail -1
done
python -m pytest tests/test_fused_bins_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "das" 2>&1 | tail -2
