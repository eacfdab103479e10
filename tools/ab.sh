#!/bin/bash
# Same-box A/B of the working tree's kernel library against gpurun_dbg/libimmunostruct_hip_ref.so (tools/build_rev_lib.sh):
# interleaved runs of tools/layer_ab.py and bench.py (iedb twice, paired + stress once), plus the layer-level GPU tests.
#   gpurun -- 'bash tools/ab.sh [pytest -k expression]'
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
L0=gpurun_dbg/libimmunostruct_hip_ref.so
K=${1:-egnn or full_train_step or golden or gather}
for lib in $L0 "" $L0 ""; do IMMUNOSTRUCT_LIB=$lib python tools/layer_ab.py "lib=$lib" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(('ref' if d['label'][4:] else 'new'), d['kernels_us']['egnn_layer_fwd'], d['kernels_us']['egnn_layer_bwd'], d['grad_digest'][:2])"; done
python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -2
for lib in $L0 "" $L0 ""; do IMMUNOSTRUCT_LIB=$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('iedb', 'ref' if '$lib' else 'new',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd',r['fwd']['slot']['mean'],r['fwd']['span']['mean'],'bwd',r['bwd']['slot']['mean'],r['bwd']['span']['mean'])"; done
for lib in $L0 ""; do for w in paired stress; do IMMUNOSTRUCT_LIB=$lib python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('$w', 'ref' if '$lib' else 'new',d['value'],d['ms_per_step'],'fwd',r['fwd']['slot']['mean'],'bwd',r['bwd']['slot']['mean'])"; done; done
