#!/bin/bash
# the paired 512-thread backward kernel against the 256-thread one (plain tiles in both), at the level the choice is made at: the
# replayed step of every bench workload, interleaved on one box.   gpurun -- 'bash tools/exp_density_paired.sh'
cd ${GRAFT_REPO_ROOT:-.}; export PYTHONPATH=$PWD
run() { IMMUNOSTRUCT_BWD_PAIRED=$1 python bench.py --workload $2 --steps 30 --warmup 5 --no-e2e --no-cpu-baseline --no-copy-ceiling 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline']['insitu_us'];print('paired=$1 $2',d['value'],d['ms_per_step'],d['step_ms']['median'],'bwd slot',r['bwd']['slot']['mean'])"; }
for i in 1 2; do for w in iedb paired; do for p in 1 0; do run $p $w; done; done; done
