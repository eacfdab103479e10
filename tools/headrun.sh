# same-box A/B of the fusion head's fp64 sums: this build against gpurun_dbg/libimmunostruct_hip_oldhead.so (built from the commit before)
run() { env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for rep in 1 2 3; do run A=new; run IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_oldhead.so; done
