cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for A in "--algo lcmv --mics 16 --frames 32768" "--algo mvdr --mics 16 --frames 32768"; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py $A --iters 20 | tail -1
echo -n "new  "; python tools/run_das.py $A --iters 20 | tail -1
done; done
echo -n "new mvdr16 2-wave "; BF_COV2D=2 python tools/run_das.py --algo mvdr --mics 16 --frames 32768 --iters 20 | tail -1
python -m pytest tests/test_pipeline_gpu.py tests/test_variants_gpu.py tests/test_shard_gpu.py -x -q -m gpu 2>&1 | tail -3
