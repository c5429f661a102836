cd /root/repo
export TMPDIR=/tmp
for seed in 11 12 13; do timeout 1500 python tools/fuzz_parity.py $seed 500 > gpurun_out/fuzz_$seed.log 2>&1; tail -4 gpurun_out/fuzz_$seed.log; done
