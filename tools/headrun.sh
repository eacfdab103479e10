# same-box A/B of the fusion head's fp64 sums: this build against gpurun_dbg/libimmunostruct_hip_oldhead.so (the previous commit's)
python tests/tools/grad_error_probe_model.py 2>&1 | grep -E "final_output|dL/d x_gat|dL/dh_L|<--|combined_attention.w"; python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu -k "comb or classifier or head or golden or mlp or full_train" 2>&1 | tail -3
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d.get('kernel_timers_us',{});print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'],{a:b for a,b in k.items() if 'comb' in a})"; }
for rep in 1 2 3; do run A=new; run IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_oldhead.so; done
for rep in 1 2; do
for l in new old; do L=""; [ $l = old ] && L="IMMUNOSTRUCT_LIB=gpurun_dbg/libimmunostruct_hip_oldhead.so"; env $L A=$l python bench.py --workload paired --steps 30 --warmup 5 --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('paired $l',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done; done
