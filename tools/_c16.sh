mkdir -p gpurun_out/c16
python -m pytest tests -m gpu -x -q -k "captured or adam or trajectory or train_model_device or entry_scripts_run" > gpurun_out/c16/pytest.log 2>&1; tail -3 gpurun_out/c16/pytest.log; grep -n "^E  " gpurun_out/c16/pytest.log | head -5
for v in 0 1 0 1; do IMMUNOSTRUCT_ADAM_OVERLAP=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c16/bench_$v.json 2>> gpurun_out/c16/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c16/bench_$v.json'));print('adam_overlap=$v',d['value'],d['ms_per_step'],d['step_ms']['median'], d['config']['final_loss'])"; done
for v in 0 1; do IMMUNOSTRUCT_ADAM_OVERLAP=$v python bench.py --workload paired --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c16/benchp_$v.json 2>> gpurun_out/c16/bench.err; python -c "
import json;d=json.load(open('gpurun_out/c16/benchp_$v.json'));print('paired adam_overlap=$v',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
tail -3 gpurun_out/c16/bench.err
