cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('auto: iedb',d['value'],d['ms_per_step'],d['step_ms']['median'],'e2e',d['e2e']['value'])"
python bench.py --workload paired --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('auto: paired',d['value'],d['ms_per_step'],d['step_ms']['median'])"
python -m pytest tests -m gpu -x -q -k "captured or entry or trajectory or distributed" 2>&1 | tail -2
