mkdir -p gpurun_out/c5
python tools/_dbg_heads.py > gpurun_out/c5/dbg_heads.txt 2>&1; cat gpurun_out/c5/dbg_heads.txt | grep -v Warn | head -40
python -m pytest tests/test_gpu_models.py tests/test_gpu_distributed.py -m gpu -x -q -k "captured or distributed or ranks or dp or bench" > gpurun_out/c5/pytest.log 2>&1; tail -3 gpurun_out/c5/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c5/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'bwd', r['mean_launch_us'], 'eager', r['eager_us'])
print(json.dumps(r.get('insitu_us')))
PY
python tools/dp_overlap_emulation.py --channels 16,32 --reserved 0,16,32 0 150 300 450 > gpurun_out/c5/dp_emulation.jsonl 2> gpurun_out/c5/dp_emulation.err; cat gpurun_out/c5/dp_emulation.jsonl; tail -3 gpurun_out/c5/dp_emulation.err
