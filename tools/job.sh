cd /root/repo
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -3 gpurun_out/bench_now.err
python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -2
