cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
echo "== block"; BF_HOP_SPIN=0 python tools/hop_latency.py 2>&1 | grep -v amdgpu
echo "== spin"; python tools/hop_latency.py 2>&1 | grep -v amdgpu
done
