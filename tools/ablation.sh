#!/bin/bash
# on the GPU box: per-launch time of the two layer kernels under every ablation build (tools/build_ablation_libs.sh), interleaved
# with the unmodified build, twice; B / DEG as tools/layer_ab.py.   bash tools/ablation.sh [out.txt]
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
out=${1:-gpurun_out/ablation.txt}
: > $out
for rep in 1 2; do
  for n in base silu seg stores nomfma nonode; do
    lib=gpurun_dbg/libimmunostruct_hip_abl_$n.so
    [ -f $lib ] || continue
    IMMUNOSTRUCT_LIB=$lib python tools/layer_ab.py $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']
print('%-7s fwd %6.2f  fwd_nocoord %6.2f  bwd %6.2f  bwd_nocoord %6.2f  eager_step_ms %.3f' % (d['label'], k.get('egnn_layer_fwd',0), k.get('egnn_layer_fwd_nocoord',0), k.get('egnn_layer_bwd',0), k.get('egnn_layer_bwd_nocoord',0), d['eager_step_ms']))" | tee -a $out
  done
done
