cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo mcra --mics 1 --streams 256 --frames 256 --iters 20 | tail -1
echo -n "new  "; python tools/run_das.py --algo mcra --mics 1 --streams 256 --frames 256 --iters 20 | tail -1
done
python -m pytest tests/test_pipeline_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "mcra" 2>&1 | tail -2
