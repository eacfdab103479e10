#!/bin/bash
# one workgroup per CU against two (round 5): per-launch time of the two layer kernels and the backward's stage stamps on the
# 256-workgroup grid (IMMUNOSTRUCT_RESERVED_CUS=128, plain 16-node tiles) and on the full one.
#   gpurun -- 'bash tools/exp_grid.sh'
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD
for r in 0 128 0 128; do
  IMMUNOSTRUCT_RESERVED_CUS=$r IMMUNOSTRUCT_BWD_TILES=0 python tools/layer_ab.py "reserved=$r" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], d['kernels_us'], d['eager_step_ms'])"
done
for r in 0 128; do
  echo "== stamps, reserved=$r"
  IMMUNOSTRUCT_RESERVED_CUS=$r IMMUNOSTRUCT_BWD_TILES=0 python tools/bwd_stamps.py 2>&1 | tail -8
done
