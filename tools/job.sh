cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_pipeline_gpu.py tests/test_fused_bins_gpu.py tests/test_golden_gpu.py tests/test_dirs_gpu.py tests/test_hops_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/run_das.py --algo phasempf --streams 256 --frames 256 --iters 10 | tail -1
python tools/run_das.py --algo phase --iters 10 | tail -1
python - <<'PY'
import sys, torch
sys.path.insert(0,'/root/repo')
from beamform_amd.capi import Beamformer
from beamform_amd.params import make_params
M,F=8,65536
p=make_params('phase',n_mics=M)
x=(torch.rand((M,F*512),device='cuda')-0.5)*8.0
y=torch.empty(F*512,device='cuda')
bf=Beamformer(p)
s=torch.cuda.current_stream().cuda_stream
for _ in range(10): bf.process_device(x.data_ptr(),F,y.data_ptr(),0,s)
torch.cuda.synchronize()
print('phase gate open', min(bf.time_device(x.data_ptr(),F,y.data_ptr(),5,s)[0] for _ in range(3)))
PY
