cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profk
rocprofv3 --kernel-trace -d /tmp/profk -o rk -- python3 /root/repo/tools/paired_step_time.py 128 merged > /tmp/profk.log 2>&1
db=$(find /tmp/profk -name "*.db" | head -1)
python3 /root/repo/tools/rocpd_timeline.py $db --step -2 > /root/repo/gpurun_out/timeline_paired.txt
