#!/bin/bash
# Same-box A/B of the working tree's kernel library against gpurun_dbg/libimmunostruct_hip_ref.so (tools/build_rev_lib.sh HEAD ref or
# tools/build_variant_lib.sh ref "<flags>"): layer timings (both backward forms), the layer-level GPU tests, interleaved bench lines.
#   gpurun -- 'bash tools/ab_lib.sh [quick]'
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
L0=gpurun_dbg/libimmunostruct_hip_ref.so
run() { IMMUNOSTRUCT_LIB=$1 IMMUNOSTRUCT_BWD_PAIRED=$2 python tools/layer_ab.py "$3" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['label'], 'fwd', k['egnn_layer_fwd'], 'bwd', k['egnn_layer_bwd'], 'red', k['reduce_partials_batched'], d['eager_step_ms'], d['grad_digest'][:3])"; }
for i in 1 2; do run $L0 1 ref-paired; run "" 1 new-paired; run $L0 0 ref-256; run "" 0 new-256; done
[ "$1" == "quick" ] && exit 0
python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or golden or deterministic or reference_default or stress" 2>&1 | tail -3
IMMUNOSTRUCT_BWD_PAIRED=0 python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or deterministic" 2>&1 | tail -2
bench() { IMMUNOSTRUCT_LIB=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e $3 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('$2 $3',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd',r['fwd']['slot']['mean'],'bwd',r['bwd']['slot']['mean'],r['bwd']['span']['mean'])"; }
for i in 1 2 3; do bench $L0 ref; bench "" new; done
for w in paired stress; do bench $L0 ref "--workload $w"; bench "" new "--workload $w"; done
