cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do
for n in default prio1 prio2 prio3; do
  if [ $n = default ]; then unset BFCORE_LIB; else export BFCORE_LIB=/root/repo/abtmp/libbf_$n.so; fi
  echo -n "$n: "; python tools/run_das.py --algo das --iters 50 --settle-ms 150 | tail -1 | cut -c28-80
done; done
