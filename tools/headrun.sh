# same-box A/B: the fusion head's workgroups alone on their CUs (-DCA_EXCLUSIVE=1 build) against the tree's build
run() { env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combined_attention" 2>&1 | tail -1
IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_caexcl.so python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "combined_attention" 2>&1 | tail -1
for rep in 1 2 3; do run A=tree; run IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_caexcl.so; done
for l in tree excl; do L="A=1"; [ $l = excl ] && L="IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_caexcl.so"; env $L python bench.py --workload paired --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('paired $l',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
