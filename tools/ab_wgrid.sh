# node weight-gradient grids re-swept under the fixed harness (IMMUNOSTRUCT_WGRAD_GRID_NODE / _PROJ; default 56 : 56)
run() { env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for rep in 1 2; do
  run IMMUNOSTRUCT_WGRAD_GRID_NODE=56 IMMUNOSTRUCT_WGRAD_GRID_PROJ=56
  run IMMUNOSTRUCT_WGRAD_GRID_NODE=64 IMMUNOSTRUCT_WGRAD_GRID_PROJ=64
  run IMMUNOSTRUCT_WGRAD_GRID_NODE=64 IMMUNOSTRUCT_WGRAD_GRID_PROJ=48
  run IMMUNOSTRUCT_WGRAD_GRID_NODE=72 IMMUNOSTRUCT_WGRAD_GRID_PROJ=56
  run IMMUNOSTRUCT_WGRAD_GRID_NODE=60 IMMUNOSTRUCT_WGRAD_GRID_PROJ=60
done
