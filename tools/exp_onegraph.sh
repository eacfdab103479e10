#!/bin/bash
# the data-parallel step as ONE captured graph (RCCL's collectives inside) under a one-rank nccl group: parity check, then the
# bench line with the form forced off / chosen by timing
cd ${GRAFT_REPO_ROOT:-.}
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
[ "$1" == "check" ] && IMMUNOSTRUCT_FORCE_COLLECTIVE=1 MASTER_PORT=29580 timeout 600 python tools/dp_rccl1_check.py 2>&1 | grep -v Warning | tail -12
for og in 0 auto 0 auto; do
IMMUNOSTRUCT_DP_ONE_GRAPH=$og IMMUNOSTRUCT_FORCE_COLLECTIVE=1 MASTER_PORT=$((29600 + RANDOM % 200)) timeout 600 python bench.py --force-pack --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-kernel-timers 2>gpurun_out/r05_e5_err_$og.txt | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());g=d['config']['grad_allreduce'];print('one_graph=$og',d['value'],d['ms_per_step'],d['step_ms']['median'],g['form'],g['one_graph_error'],g['tuned_ms'])"
done
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-kernel-timers 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('single',d['value'],d['ms_per_step'],d['step_ms']['median'])"
