python -m pytest tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do IMMUNOSTRUCT_FORCE_COLLECTIVE=1 MASTER_PORT=2959$rep python bench.py --force-pack --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('rccl1',d['value'],d['ms_per_step'],d['step_ms']['median'],d['config']['grad_allreduce']['form'],d['config']['grad_allreduce']['tuned_ms'])"; done
bash tools/rccl1_timeline.sh | tail -22
