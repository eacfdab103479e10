mkdir -p gpurun_out/c12
for c in 0 1 0 1; do B=256 CLOCKS=$c python tools/layer_ab.py "clocks=$c" >> gpurun_out/c12/ab.jsonl 2>> gpurun_out/c12/ab.err; done
python - <<'PY'
import json
for l in open('gpurun_out/c12/ab.jsonl'):
    d=json.loads(l); print(d['label'], d['kernels_us'].get('egnn_layer_fwd'), d['kernels_us'].get('egnn_layer_bwd'), d['eager_step_ms'])
PY
for f in "" "--no-kernel-timers"; do python bench.py --workload paired --steps 20 --warmup 5 --no-cpu-baseline $f > gpurun_out/c12/b.json 2>> gpurun_out/c12/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c12/b.json'));print('paired [$f]',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
