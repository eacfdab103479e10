#!/bin/bash
# the backward layer launch over batch sizes: plain 16-node tiles (the paired 512-thread kernel) against the greedy <= 64-edge tile list
# (the 256-thread kernel) -- the measurement behind functional.use_bwd_tiles.   gpurun -- 'bash tools/exp_tiles_sweep.sh'
cd ${GRAFT_REPO_ROOT:-.}; export PYTHONPATH=$PWD
for b in ${@:-128 150 192 256 384 512}; do for t in plain listed; do
  B=$b TILES=$t python tools/layer_ab.py "B=$b $t" 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_us']; print(d['label'], 'E', d['E'], 'bwd', k['egnn_layer_bwd'], 'fwd', k['egnn_layer_fwd'], 'red', k.get('reduce_partials_batched'), 'stack eager ms', d['eager_step_ms'])"
done; done
