mkdir -p gpurun_out/c9
python -m pytest tests -m gpu -x -q -k "full_train_step or paired_product or reduce or egnn_layer or egnn_stack or trajectory" > gpurun_out/c9/pytest.log 2>&1; tail -3 gpurun_out/c9/pytest.log; grep -n "^E  " gpurun_out/c9/pytest.log | head -3
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c9/bench.json 2>> gpurun_out/c9/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c9/bench.json'));print(d['value'],d['ms_per_step'],d['step_ms']['median'], {k:v[1] for k,v in d['kernel_timers_us'].items() if 'reduce' in k})"
