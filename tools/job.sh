cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "phasempf" 2>&1 | tail -2
python -m pytest tests/test_hops_gpu.py tests/test_fused_bins_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo phasempf --streams 256 --frames 256 --iters 20 | tail -1
echo -n "new  "; python tools/run_das.py --algo phasempf --streams 256 --frames 256 --iters 20 | tail -1
done
