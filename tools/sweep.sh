#!/bin/bash
# usage: tools/sweep.sh ENVVAR v1 v2 ... -- runs the default bench for every value and prints ms/step
var=$1; shift
for v in "$@"; do
  out=$(env $var=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timers 2>/dev/null | tail -1)
  echo "$var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
