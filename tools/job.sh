cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -2 gpurun_out/bench_final.err
