cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3; do python -m pytest tests -x -q -m gpu 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()"
