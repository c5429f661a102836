cd /root/repo
export TMPDIR=/tmp
for sd in 61 62 63 64 65 66 67 68; do timeout 600 python tools/fuzz_parity.py $sd 250 2>&1 | tail -1; done
