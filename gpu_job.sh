cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fused_bins_gpu.py tests/test_variants_gpu.py tests/test_das_gpu.py tests/test_pipeline_gpu.py tests/test_edges_gpu.py -q > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
timeout 300 python bench.py --no-cpu --no-extra > gpurun_out/t.json 2>/dev/null; python -c "
import json
d=json.loads([l for l in open('gpurun_out/t.json') if l.startswith('{')][0]); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
