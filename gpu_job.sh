cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/full_gpu.log 2>&1; tail -6 gpurun_out/full_gpu.log
timeout 900 python bench.py > gpurun_out/r04_c_bench_das8.json 2> gpurun_out/r04_c_bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_c_bench_das8.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
for k,v in d['extra'].items():
    if isinstance(v,dict): print(k, v.get('ms_per_step'))
PY
