cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['extra'], d['cpu_baseline']['value'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_e -- python bench.py --no-cpu --no-extra > gpurun_out/prof_e.log 2>&1
for f in $(find gpurun_out/prof_e -name "*kernel_stats*"); do cut -c1-160 $f | head -3; done
