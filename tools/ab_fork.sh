# where the sequence branch is forked from the EGNN forward stack (IMMUNOSTRUCT_FORK_AFTER_LAYER), interleaved bench lines on one box
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for rep in 1 2; do for k in 1 2 3 4 5; do run IMMUNOSTRUCT_FORK_AFTER_LAYER=$k; done; done
for k in 1 2 3 4; do IMMUNOSTRUCT_FORK_AFTER_LAYER=$k python bench.py --workload paired --steps 30 --warmup 5 --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('paired fork=$k',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
