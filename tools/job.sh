cd /root/repo
export TMPDIR=/tmp
for seed in 11 12 13 14; do python tools/fuzz_parity.py $seed 250 2>&1 | tail -3; done
