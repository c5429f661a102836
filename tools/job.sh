cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo mvdr --iters 30 | tail -1
echo -n "new  "; python tools/run_das.py --algo mvdr --iters 30 | tail -1
done
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -2
