# A/B of the deferred loss (IMMUNOSTRUCT_DEFER_LOSS, IMMUNOSTRUCT_DEFER_TOTAL) on one box: interleaved bench lines
mkdir -p gpurun_out/defer
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "speculative or captured_hip or trajectory or golden" 2>&1 | tail -3
IMMUNOSTRUCT_DEFER_TOTAL=caller python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "captured_hip or training_trajectory" 2>&1 | tail -3
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for rep in 1 2 3; do
  run IMMUNOSTRUCT_DEFER_LOSS=0
  run IMMUNOSTRUCT_DEFER_LOSS=1
  run IMMUNOSTRUCT_DEFER_LOSS=1 IMMUNOSTRUCT_DEFER_TOTAL=caller
  run IMMUNOSTRUCT_DEFER_LOSS=0 IMMUNOSTRUCT_SIDE_PRIORITY=1
  run IMMUNOSTRUCT_DEFER_LOSS=1 IMMUNOSTRUCT_DEFER_TOTAL=caller IMMUNOSTRUCT_SIDE_PRIORITY=1
done
export TMPDIR=/tmp
IMMUNOSTRUCT_DEFER_TOTAL=caller rocprofv3 --kernel-trace -d /tmp/prof_d -o rr -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e --no-copy-ceiling > gpurun_out/defer/prof.log 2> gpurun_out/defer/prof.err
db=$(find /tmp/prof_d -name "*.db" | head -1); python tools/rocpd_timeline.py $db > gpurun_out/defer/timeline_caller.txt; sed -n 18,55p gpurun_out/defer/timeline_caller.txt
