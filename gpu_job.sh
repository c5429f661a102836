cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_bins_gpu.py tests/test_variants_gpu.py -q -k "one_launch or every_one_launch" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
for i in 1 2; do
timeout 120 python tools/time_das_f64.py 8 65536 > gpurun_out/t.log 2>&1; echo "lds  $(grep kernel-only gpurun_out/t.log)"
BFCORE_LIB=/root/repo/beamform_amd/lib/libbfcore_swaps.so timeout 120 python tools/time_das_f64.py 8 65536 > gpurun_out/t.log 2>&1; echo "swap $(grep kernel-only gpurun_out/t.log)"
done
