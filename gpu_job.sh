cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for v in "" _fsl9 _fsl11; do
  if [ -z "$v" ]; then lib=/root/repo/beamform_amd/lib/libbfcore.so; else lib=/root/repo/beamform_amd/lib/libbfcore$v.so; fi
  BFCORE_LIB=$lib timeout 120 python tools/run_das.py --algo das --iters 20 > gpurun_out/t.log 2>&1; echo "planar '$v' $(tail -1 gpurun_out/t.log | cut -c1-70)"
  BFCORE_LIB=$lib timeout 120 python tools/run_das.py --algo das --layout interleaved --iters 20 > gpurun_out/t.log 2>&1; echo "interl '$v' $(tail -1 gpurun_out/t.log | cut -c1-70)"
done; done
