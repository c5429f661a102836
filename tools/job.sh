cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -4 gpurun_out/bench_now.err
