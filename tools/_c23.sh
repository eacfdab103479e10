L0=gpurun_dbg/libimmunostruct_hip_oldbwd.so
for lib in $L0 "" $L0 ""; do IMMUNOSTRUCT_LIB=$lib python tools/layer_ab.py "lib=$lib" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print((d['label'][-9:-3] or "new"), d['kernels_us']['egnn_layer_fwd'], d['kernels_us']['egnn_layer_bwd'], d['grad_digest'][:2])"; done
python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or golden" 2>&1 | tail -2
for lib in $L0 "" $L0 ""; do IMMUNOSTRUCT_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('lib=[$lib]',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd',r['fwd']['slot']['mean'],r['fwd']['span']['mean'],'bwd',r['bwd']['slot']['mean'])"; done
for lib in $L0 ""; do for w in paired stress; do IMMUNOSTRUCT_LIB=$lib python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('$w lib=[$lib]',d['value'],d['ms_per_step'],'fwd',r['fwd']['slot']['mean'],'bwd',r['bwd']['slot']['mean'])"; done; done
