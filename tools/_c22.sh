mkdir -p gpurun_out/c22
python -m pytest tests -m gpu -x -q > gpurun_out/c22/pytest.log 2>&1; tail -3 gpurun_out/c22/pytest.log; grep -n "^E  " gpurun_out/c22/pytest.log | head -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/round_profiles.sh r03b > gpurun_out/c22/round.log 2>&1; tail -6 gpurun_out/c22/round.log | cut -c1-200
