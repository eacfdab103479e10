cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
bash tools/round_profiles.sh r03c
bash tools/pmc_passes.sh gpurun_out/r03c/sq_counters_layer_bwd.txt egnn_layer_bwd
bash tools/pmc_passes.sh gpurun_out/r03c/sq_counters_layer_fwd.txt egnn_layer_fwd
bash tools/pmc_passes.sh gpurun_out/r03c/sq_counters_node_wgrad.txt wgrad16_batched
python tools/wg_clock_profile.py > gpurun_out/r03c/wg_clock_profile.txt 2>&1
