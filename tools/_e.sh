cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD
for lib in "" gpurun_dbg/libimmunostruct_hip_nocls.so "" gpurun_dbg/libimmunostruct_hip_nocls.so; do IMMUNOSTRUCT_LIB=$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());t=d['kernel_timers_us'];print('lib=[$lib]', d['ms_per_step'], 'comb bwd', t['comb_attn_cls_bwd'][1], 'comb fwd', t['comb_attn_cls_fwd'][1])"; done
