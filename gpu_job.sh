cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_bins_gpu.py -x -q -k "das_f64" 2>&1 | tail -4
timeout 120 python - <<'PY'
import sys; sys.path.insert(0,'.')
import torch, numpy as np
from beamform_amd.capi import BF_DAS_BINS_F64, BF_INTERLEAVED, Beamformer
from beamform_amd.params import make_params
M,F=8,65536
x=torch.rand((F*512,M),device='cuda')-0.5
y=torch.empty(F*512,device='cuda')
bf=Beamformer(make_params('das',n_mics=M),das_impl=BF_DAS_BINS_F64,layout=BF_INTERLEAVED)
s=torch.cuda.current_stream().cuda_stream
for _ in range(30): bf.process_device(x.data_ptr(),F,y.data_ptr(),0,s)
torch.cuda.synchronize()
print('f64 interleaved', min(bf.time_device(x.data_ptr(),F,y.data_ptr(),10,s)[0] for _ in range(4)))
PY
timeout 120 python tools/time_das_f64.py | tail -1
