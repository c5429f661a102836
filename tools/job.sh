cd /root/repo
python tools/time_lcmv.py 2>&1 | grep -v amdgpu
