mkdir -p gpurun_out/c11
for lib in "" gpurun_dbg/libimmunostruct_hip_noclk.so "" gpurun_dbg/libimmunostruct_hip_noclk.so; do B=256 IMMUNOSTRUCT_LIB=$lib python tools/layer_ab.py "lib=$lib" >> gpurun_out/c11/ab.jsonl 2>> gpurun_out/c11/ab.err; done
python - <<'PY'
import json
for l in open('gpurun_out/c11/ab.jsonl'):
    d=json.loads(l); print(d['label'][-30:], d['E'], d['kernels_us'].get('egnn_layer_fwd'), d['kernels_us'].get('egnn_layer_bwd'), d['eager_step_ms'])
PY
IMMUNOSTRUCT_BENCH_WORKLOAD=paired python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c11/bench_paired.json 2>> gpurun_out/c11/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c11/bench_paired.json'));print('paired',d['value'],d['ms_per_step'],d['kernel_timers_us']['egnn_layer_bwd'])"
