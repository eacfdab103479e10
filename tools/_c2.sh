mkdir -p gpurun_out/c2
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -x -q -k "egnn or full_train_step or head_counts or trajectory or golden" > gpurun_out/c2/pytest.log 2>&1; tail -4 gpurun_out/c2/pytest.log
for v in 1 0 1 0; do IMMUNOSTRUCT_SAVE_Z3=$v python tools/layer_ab.py save_z3=$v >> gpurun_out/c2/ab.jsonl 2>> gpurun_out/c2/ab.err; done
cat gpurun_out/c2/ab.jsonl
for v in 1 0 1 0; do IMMUNOSTRUCT_SAVE_Z3=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-kernel-timers > gpurun_out/c2/bench_$v.json 2>> gpurun_out/c2/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c2/bench_$v.json'));print('save_z3=$v',d['value'],d['ms_per_step'],d.get('step_ms'))"; done
