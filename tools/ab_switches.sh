# the placement switches once more, under the harness that no longer times a clock ramp (HISTORY 7.15): interleaved bench lines
run() { env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
for rep in 1 2; do
  run BASE=1
  run IMMUNOSTRUCT_DEFER_LOSS=1
  run IMMUNOSTRUCT_ADAM_OVERLAP=1 IMMUNOSTRUCT_ADAM_GATE_LAYER=0
  run IMMUNOSTRUCT_ADAM_OVERLAP=1 IMMUNOSTRUCT_ADAM_GATE_LAYER=2
  run IMMUNOSTRUCT_ADAM_EARLY_PREPARE=1
  run IMMUNOSTRUCT_STEP_RANDOM=torch
  run IMMUNOSTRUCT_EARLY_JOIN=0
done
