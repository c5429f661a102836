cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for n in default w0p1 w1p0 w0p0; do
  if [ $n = default ]; then unset BFCORE_LIB; else export BFCORE_LIB=/root/repo/abtmp/libbf_$n.so; fi
  echo "== $n"; python tools/time_scene.py mvdr 8 65536 4 > /tmp/o.txt 2>&1; head -2 /tmp/o.txt
done; done
