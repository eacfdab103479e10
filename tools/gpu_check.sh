#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/gpu_check.sh TAG [pytest-args...]
#   runs the GPU tests (or the given selection), the default bench line, and a rocprofv3 kernel trace of a short bench
#   whose per-kernel table and last-step timeline land in gpurun_out/TAG/
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
if [ "$#" -gt 0 ]; then sel="$@"; else sel="tests -m gpu"; fi
python -m pytest $sel -x -q 2>&1 | tail -40 > $out/tests.log
tail -3 $out/tests.log
python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
cut -c1-400 $out/bench.json
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o rr -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e > $out/prof.log 2> $out/prof.err
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $out/kernel_stats.txt 2>> $out/prof.err
python tools/rocpd_timeline.py $db > $out/timeline.txt 2>> $out/prof.err
tail -60 $out/timeline.txt
