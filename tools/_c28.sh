cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or golden or gather" 2>&1 | tail -2
for g in "56 56" "56 56" "64 64" "48 48"; do set -- $g; IMMUNOSTRUCT_WGRAD_GRID_NODE=$1 IMMUNOSTRUCT_WGRAD_GRID_PROJ=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());t=d['kernel_timers_us'];print('grids $g', d['ms_per_step'],d['step_ms']['median'],'wgrad',t['egnn_node_wgrad_batched'][1],'reduce',t['reduce_partials_batched'][1])"; done
