cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_fused_bins_gpu.py -x -q 2>&1 | tail -5
timeout 600 python bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; tail -3 gpurun_out/r4c/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4c/bench.json'))
print(d['dtype'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['kernel_ms'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('error'), (v.get('roofline') or {}).get('frac'))
PY
