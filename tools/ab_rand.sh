# A/B of engine.CapturedTrainStep(prefetch_random) on one box (IMMUNOSTRUCT_PREFETCH_RANDOM), interleaved bench lines + a timeline
mkdir -p gpurun_out/rand
python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q -m gpu -k "prefetched or captured or trajectory or device or step_random" 2>&1 | tail -3
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'],'e2e',(d.get('e2e') or {}).get('value'))"; }
for rep in 1 2 3; do
  run IMMUNOSTRUCT_STEP_RANDOM=torch
  run IMMUNOSTRUCT_STEP_RANDOM=device
done
for p in torch device; do IMMUNOSTRUCT_STEP_RANDOM=$p python bench.py --workload paired --steps 30 --warmup 5 --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('paired prefetch=$p',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_r -o rr -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e --no-copy-ceiling > gpurun_out/rand/prof.log 2> gpurun_out/rand/prof.err
db=$(find /tmp/prof_r -name "*.db" | head -1); python tools/rocpd_timeline.py $db > gpurun_out/rand/timeline.txt; sed -n 1,30p gpurun_out/rand/timeline.txt
