cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
timeout 300 python tools/bwd_stamps.py 2>&1 | tail -9 | grep -v "first tile" | cut -c1-250
bash tools/ab.sh
