#!/bin/bash
# early staging of the backward layer kernel's weight tiles (round 5) against the build without it (gpurun_dbg/..._ref.so =
# tools/build_variant_lib.sh ref -DIS_BWD_EARLY_STAGE=0), and the paired 512-thread kernel, same box, interleaved
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
L0=gpurun_dbg/libimmunostruct_hip_ref.so
run() { IMMUNOSTRUCT_LIB=$1 IMMUNOSTRUCT_BWD_PAIRED=$2 python tools/layer_ab.py "$3" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['label'], 'fwd', k['egnn_layer_fwd'], 'bwd', k['egnn_layer_bwd'], 'red', k['reduce_partials_batched'], d['eager_step_ms'], d['grad_digest'][:2])"; }
for i in 1 2; do run $L0 0 ref; run "" 0 early; run "" 1 paired; done
echo "== stamps early"; IMMUNOSTRUCT_BWD_PAIRED=0 python tools/bwd_stamps.py 2>&1 | tail -4
[ "$1" == "quick" ] && exit 0
IMMUNOSTRUCT_BWD_PAIRED=0 python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or golden or deterministic or reference_default or stress" 2>&1 | tail -3
bench() { IMMUNOSTRUCT_LIB=$1 IMMUNOSTRUCT_BWD_PAIRED=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e $4 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('$3 $4',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd',r['fwd']['slot']['mean'],'bwd',r['bwd']['slot']['mean'],r['bwd']['span']['mean'])"; }
for i in 1 2; do bench $L0 0 ref; bench "" 0 early; bench "" 1 paired; done
for w in paired stress; do bench $L0 0 ref "--workload $w"; bench "" 0 early "--workload $w"; bench "" 1 paired "--workload $w"; done
