#!/bin/bash
# usage: tools/pmc_traffic.sh <workload> <out.json>   -- two PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py --eager for that workload
wl=$1; out=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python3 $root/bench.py --workload $wl --eager --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-e2e > /tmp/pmc_$c.log 2>&1
done
python3 $root/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1) > $root/$out
