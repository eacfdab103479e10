#!/bin/bash
# usage (on the GPU box): bash tools/gpu_stack_prof.sh TAG [pytest selection...]
#   optional test selection first, then a rocprofv3 kernel trace of tools/edge_stack_run.py (3 x forward + backward of the
#   6-layer EGNN stack at B = 128): per-kernel table in gpurun_out/TAG/stack_stats.txt
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
if [ "$#" -gt 0 ]; then
  python -m pytest "$@" -x -q 2>&1 | tail -40 > $out/tests.log
  tail -4 $out/tests.log
fi
export TMPDIR=/tmp
rm -rf /tmp/sprof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/sprof_$tag -o rr -- python3 tools/edge_stack_run.py > $out/sprof.log 2> $out/sprof.err
db=$(find /tmp/sprof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $out/stack_stats.txt 2>> $out/sprof.err
head -12 $out/stack_stats.txt | cut -c1-150
