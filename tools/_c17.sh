mkdir -p gpurun_out/c17
export TMPDIR=/tmp
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o rr -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers --no-e2e > gpurun_out/c17/prof.log 2> gpurun_out/c17/prof.err
db=$(find /tmp/prof_t -name "*.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/c17/timeline.txt 2>> gpurun_out/c17/prof.err
tail -22 gpurun_out/c17/timeline.txt
