cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profp
rocprofv3 --kernel-trace --stats -d /tmp/profp -o rp -- python3 /root/repo/tools/paired_step_time.py 128 merged > /tmp/profp.log 2>&1
db=$(find /tmp/profp -name "*.db" | head -1)
python3 /root/repo/tools/rocpd_stats.py $db --top 40 > /root/repo/gpurun_out/paired_merged_stats.txt
python3 /root/repo/tools/rocpd_timeline.py $db --step -2 > /root/repo/gpurun_out/timeline_paired.txt
