cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for A in "--algo mvdr" "--algo lcmv --mics 16 --frames 32768" "--algo lcmv --mics 8"; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py $A --iters 20 | tail -1
echo -n "nr1  "; python tools/run_das.py $A --iters 20 | tail -1
done; done
python -m pytest tests/test_pipeline_gpu.py tests/test_variants_gpu.py tests/test_shard_gpu.py tests/test_hops_gpu.py -x -q -m gpu 2>&1 | tail -2
python tools/fuzz_parity.py 41 200 2>&1 | tail -1
