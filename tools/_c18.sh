for v in 0 1 0 1; do IMMUNOSTRUCT_ADAM_PREPARE_EARLY=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('prepare_early=$v',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
python -m pytest tests -m gpu -x -q -k "captured or trajectory or train_model_device" 2>&1 | tail -2
