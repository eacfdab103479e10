mkdir -p gpurun_out/c4
python -m pytest tests -m gpu -x -q > gpurun_out/c4/pytest.log 2>&1; tail -3 gpurun_out/c4/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c4/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['mean_launch_us'], r['eager_us'], r['traffic'], r['traffic_source'][:60], r.get('insitu_us'))
print(r['forward_kernel']['mean_launch_us'], r['forward_kernel']['eager_us'], r['gather_kernel'])
PY
python bench.py --stage finetune --steps 30 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/c4/bench_ft.json 2>> gpurun_out/c4/bench.err; cut -c1-330 gpurun_out/c4/bench_ft.json
IMMUNOSTRUCT_BENCH_WORKLOAD=paired python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c4/bench_paired.json 2>> gpurun_out/c4/bench.err; cut -c1-200 gpurun_out/c4/bench_paired.json
IMMUNOSTRUCT_BENCH_WORKLOAD=stress python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c4/bench_stress.json 2>> gpurun_out/c4/bench.err; python -c "
import json; d=json.load(open('gpurun_out/c4/bench_stress.json')); print(d['value'], d['roofline'].get('insitu_us'))"
tail -5 gpurun_out/c4/bench.err
