cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python tools/time_node.py lcmv 16 32768
python tools/time_node.py mvdr 16 32768
python tools/time_node.py lcmv 8
python tools/time_node.py gss 8
python tools/time_node.py mvdr 8 65535
python tools/time_node.py mvdr 8 1000
