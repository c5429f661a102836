cd /root/repo
export TMPDIR=/tmp
for m in 1 2 4; do echo -n "x$m: "; BF_DASF64_RUNS=$m python tools/run_das.py --algo das --das-f64 --iters 20 | tail -1 | cut -c28-60; done
for m in 1 2; do echo -n "x$m: "; BF_DASF64_RUNS=$m python tools/run_das.py --algo das --das-f64 --iters 20 | tail -1 | cut -c28-60; done
