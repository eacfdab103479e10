#!/bin/bash
# usage (on the GPU box): bash tools/paired_prof.sh   -- rocprofv3 kernel trace of the paired (cancer / wild-type) bench step:
#   gpurun_out/paired_timeline.txt = per-queue timeline of the last replayed step (tools/rocpd_timeline.py)
export TMPDIR=/tmp
rm -rf /tmp/prof_p
rocprofv3 --kernel-trace --stats -d /tmp/prof_p -o rr -- python3 bench.py --workload paired --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e > gpurun_out/paired_prof.log 2> gpurun_out/paired_prof.err
db=$(find /tmp/prof_p -name "*.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/paired_timeline.txt
python tools/rocpd_stats.py $db | head -14 > gpurun_out/paired_stats.txt
for f in 2 3 4; do echo "fork_after=$f"; IMMUNOSTRUCT_FORK_AFTER_LAYER=$f python bench.py --workload paired --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timers --no-e2e 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"ms_per_step\"], d[\"step_ms\"][\"median\"])"; done
