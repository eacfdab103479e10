mkdir -p gpurun_out/c8
python -m pytest tests -m gpu -x -q > gpurun_out/c8/pytest.log 2>&1; tail -3 gpurun_out/c8/pytest.log
bash tools/round_profiles.sh r03a > gpurun_out/c8/round.log 2>&1; tail -25 gpurun_out/c8/round.log
