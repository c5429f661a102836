cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_variants_gpu.py tests/test_hops_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "== fast"; python tools/time_lcmv.py 2>&1 | grep -v amdgpu.ids
for sd in 21 22; do python tools/fuzz_parity.py $sd 150 2>&1 | tail -3; done
