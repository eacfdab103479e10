# the step's random tensors: tests + a timeline (is::step_random_kernel's duration on the sequence branch's queue)
mkdir -p gpurun_out/rand
python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -x -q -m gpu -k "prefetched or device_side or step_random" 2>&1 | tail -2
for rep in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-copy-ceiling --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('iedb',d['value'],d['ms_per_step'],d['step_ms']['median'])"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_r -o rr -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e --no-copy-ceiling > gpurun_out/rand/prof.log 2> gpurun_out/rand/prof.err
db=$(find /tmp/prof_r -name "*.db" | head -1); python tools/rocpd_timeline.py $db > gpurun_out/rand/timeline.txt; grep -n "step_random\|span" gpurun_out/rand/timeline.txt
