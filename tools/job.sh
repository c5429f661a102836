cd /root/repo
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()"
