# kernel timeline of the data-parallel step under a one-rank RCCL group (what separates it from the single-GPU step)
mkdir -p gpurun_out/rccl1; export TMPDIR=/tmp
IMMUNOSTRUCT_FORCE_COLLECTIVE=1 MASTER_PORT=29593 IMMUNOSTRUCT_DP_OVERLAP=${DP_OVERLAP:-0} rocprofv3 --kernel-trace -d /tmp/prof_rc -o rr -- python3 bench.py --force-pack --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e --no-copy-ceiling > gpurun_out/rccl1/prof.log 2> gpurun_out/rccl1/prof.err
db=$(find /tmp/prof_rc -name "*.db" | head -1); python tools/rocpd_timeline.py $db > gpurun_out/rccl1/timeline.txt; cat gpurun_out/rccl1/timeline.txt
