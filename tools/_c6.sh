mkdir -p gpurun_out/c6
for p in flat half alt_wg alt_wave flat half; do IMMUNOSTRUCT_FWD_PATTERN=$p python tools/layer_ab.py pattern=$p >> gpurun_out/c6/ab.jsonl 2>> gpurun_out/c6/ab.err; done
python - <<'PY'
import json
for l in open('gpurun_out/c6/ab.jsonl'):
    d=json.loads(l); print(d['label'], d['kernels_us']['egnn_layer_fwd'], d['kernels_us']['egnn_layer_bwd'], d['grad_digest'][:2])
PY
for p in flat half alt_wg; do IMMUNOSTRUCT_FWD_PATTERN=$p python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c6/bench_$p.json 2>> gpurun_out/c6/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c6/bench_$p.json'));r=d['roofline']['insitu_us'];print('$p',d['value'],d['ms_per_step'],d['step_ms']['median'],'fwd slot',r['fwd']['slot']['mean'],'span',r['fwd']['span']['mean'])"; done
python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "head_counts or device_batcher" 2>&1 | tail -2
timeout 600 python tools/dp_overlap_emulation.py --channels 16,32 --reserved 0,16,32 0 150 300 450 > gpurun_out/c6/dp_emulation.jsonl 2> gpurun_out/c6/dp_emulation.err; cat gpurun_out/c6/dp_emulation.jsonl; tail -3 gpurun_out/c6/dp_emulation.err | cut -c1-300
