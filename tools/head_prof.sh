# kernel durations of the fusion head's launches alone (rocprofv3 over tools/head_time.py): usage bash tools/head_prof.sh [LIB.so ...]
export TMPDIR=/tmp
for lib in "" "$@"; do
  rm -rf /tmp/prof_h; [ -n "$lib" ] && export IMMUNOSTRUCT_LIB=$lib || unset IMMUNOSTRUCT_LIB
  rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o rr -- python3 tools/head_time.py > /dev/null 2>&1
  db=$(find /tmp/prof_h -name "*.db" | head -1); echo "lib=${lib:-tree}"; python tools/rocpd_stats.py $db | grep -E "comb_attn" | cut -c1-110
done
