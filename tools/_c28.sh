cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
timeout 300 python tools/stage_stamps.py 2>&1 | tail -3 | sed 's/.*| node half/node half/'
bash tools/ab.sh
