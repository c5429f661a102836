cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
echo -n "base "; BFCORE_LIB=/root/repo/abtmp/libbfcore_base.so python tools/run_das.py --algo gss --mics 8 --streams 256 --frames 256 --iters 10 | tail -1
echo -n "new  "; python tools/run_das.py --algo gss --mics 8 --streams 256 --frames 256 --iters 10 | tail -1
done
python -m pytest tests -x -q -m gpu -k "gss or interf or golden or dirs" 2>&1 | tail -2
python tools/fuzz_parity.py 81 300 2>&1 | tail -1
