cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for A in "--algo lcmv --mics 16 --frames 32768" "--algo mvdr --mics 16 --frames 32768"; do
echo -n "keepC  COV2D=2 "; BFCORE_LIB=/root/repo/abtmp/libbfcore_keepc.so BF_COV2D=2 python tools/run_das.py $A --iters 20 | tail -1
echo -n "keepCX COV2D=2 "; BF_COV2D=2 python tools/run_das.py $A --iters 20 | tail -1
echo -n "default        "; python tools/run_das.py $A --iters 20 | tail -1
done; done
