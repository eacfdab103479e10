# A/B of the overlapped optimizer update (IMMUNOSTRUCT_ADAM_OVERLAP=1) with the gate behind EGNN layer k's backward launch
run() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$*',d['value'],d['ms_per_step'],d['step_ms']['median'])"; }
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "captured_hip" 2>&1 | tail -2
IMMUNOSTRUCT_ADAM_OVERLAP=1 IMMUNOSTRUCT_ADAM_GATE_LAYER=2 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "captured_hip" 2>&1 | tail -2
for rep in 1 2; do
  run IMMUNOSTRUCT_ADAM_OVERLAP=0
  for k in -1 0 1 2 3 4; do run IMMUNOSTRUCT_ADAM_OVERLAP=1 IMMUNOSTRUCT_ADAM_GATE_LAYER=$k; done
done
