cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do
echo -n "base  il4 "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo das --mics 4 --layout interleaved --iters 100 | tail -1
echo -n "split il4 "; python tools/run_das.py --algo das --mics 4 --layout interleaved --iters 100 | tail -1
done
python -m pytest tests/test_das_gpu.py -x -q -m gpu 2>&1 | tail -2
