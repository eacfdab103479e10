cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r03c
timeout 900 python tools/dp_overlap_emulation.py --channels 16,32 --reserved 0,16 0 150 300 450 > gpurun_out/r03c/dp_emulation.jsonl 2> gpurun_out/r03c/dp_emulation.err
tail -3 gpurun_out/r03c/dp_emulation.jsonl | cut -c1-300
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
