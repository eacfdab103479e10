#!/bin/bash
# Same-box timing of the layer kernels under several builds of the kernel library:  gpurun -- 'bash tools/ab_libs.sh name1 name2 ...'
# (name = gpurun_dbg/libimmunostruct_hip_<name>.so, "tree" = the working tree's library); two interleaved rounds.
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
run() { local lib=""; [ "$1" != "tree" ] && lib=gpurun_dbg/libimmunostruct_hip_$1.so
  IMMUNOSTRUCT_LIB=$lib python tools/layer_ab.py "$1" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['label'], 'fwd', k['egnn_layer_fwd'], 'bwd', k['egnn_layer_bwd'], 'red', k.get('reduce_partials_batched'), d['eager_step_ms'], d['grad_digest'][:3])"; }
for i in 1 2; do for n in "$@"; do run $n; done; done
