cd /root/repo
export TMPDIR=/tmp
BFCORE_LIB=$PWD/beamform_amd/lib/libbfcore_pro.so timeout 300 python -m pytest tests/test_fused_bins_gpu.py -x -q -k das_f64 2>&1 | tail -2
for rep in 1 2 3; do for tag in base0 pro; do BFCORE_LIB=$PWD/beamform_amd/lib/libbfcore_$tag.so timeout 120 python tools/time_das_f64.py 2>&1 | grep kernel-only | sed "s/^/$tag: /"; done; done
