cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hops_gpu.py -q -m gpu 2>&1 | grep -v amdgpu | tail -3
python bench.py --no-cpu --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k in ('das_period256','das_period1024'): print(k, d['extra'][k]['ms_per_step'])
"
