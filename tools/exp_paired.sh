#!/bin/bash
# the paired backward layer kernel (one 512-thread workgroup per CU, csrc/egnn_layer_bwd8.hip) against the 256-thread one, same box:
#   gpurun -- 'bash tools/exp_paired.sh [quick]'
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
for p in 0 1 0 1; do
  IMMUNOSTRUCT_BWD_PAIRED=$p python tools/layer_ab.py "paired=$p" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], d['kernels_us'], d['eager_step_ms'], d['grad_digest'][:3])"
done
for p in 0 1; do echo "== stamps paired=$p"; IMMUNOSTRUCT_BWD_PAIRED=$p python tools/bwd_stamps.py 2>&1 | tail -4; done
[ "$1" == "quick" ] && exit 0
python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or golden or deterministic or reference_default" 2>&1 | tail -3
for p in 0 1 0 1; do IMMUNOSTRUCT_BWD_PAIRED=$p python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('iedb paired=$p',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd',r['fwd']['slot']['mean'],'bwd',r['bwd']['slot']['mean'],r['bwd']['span']['mean'])"; done
