cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu -k "loss or golden or oracle or trajectory" 2>&1 | tail -2
for m in 1 2 3; do timeout 300 python bench.py --no-kernel-timers --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_loss'])"; done
