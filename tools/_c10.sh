mkdir -p gpurun_out/c10
python -m pytest tests -m gpu -x -q -k "attention or full_train_step or golden or captured" > gpurun_out/c10/pytest.log 2>&1; tail -3 gpurun_out/c10/pytest.log; grep -n "^E  " gpurun_out/c10/pytest.log | head -3
for v in 0 1 0 1; do IMMUNOSTRUCT_ATTN_SPLIT=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c10/bench_$v.json 2>> gpurun_out/c10/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c10/bench_$v.json'));print('split=$v',d['value'],d['ms_per_step'],d['step_ms']['median'], {k:v[1] for k,v in d['kernel_timers_us'].items() if 'attn' in k})"; done
