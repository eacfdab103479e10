mkdir -p gpurun_out/c3
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "egnn or head_counts or fused_loss or ssl_losses" > gpurun_out/c3/pytest.log 2>&1; tail -3 gpurun_out/c3/pytest.log
for v in 1 0 1 0; do IMMUNOSTRUCT_SAVE_Z3=$v python tools/layer_ab.py save_z3=$v >> gpurun_out/c3/ab.jsonl 2>> gpurun_out/c3/ab.err; done
python - <<'PY'
import json
for l in open('gpurun_out/c3/ab.jsonl'):
    d=json.loads(l); print(d['label'], d['kernels_us']['egnn_layer_fwd'], d['kernels_us']['egnn_layer_bwd'], d['eager_step_ms'], d['grad_digest'][:2])
PY
for v in 1 0 1 0; do IMMUNOSTRUCT_SAVE_Z3=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-kernel-timers > gpurun_out/c3/bench_$v.json 2>> gpurun_out/c3/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c3/bench_$v.json'));print('save_z3=$v',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
