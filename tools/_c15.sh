mkdir -p gpurun_out/c15
for v in 1 0; do IMMUNOSTRUCT_SAVE_Z3=$v python tools/layer_ab.py save_z3=$v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['label'], d['kernels_us']['egnn_layer_fwd'], d['kernels_us']['egnn_layer_bwd'], d['grad_digest'][:2])"; done
python -m pytest tests -m gpu -x -q -k "egnn or full_train_step or paired_product" 2>&1 | tail -2
bash tools/_c14.sh
