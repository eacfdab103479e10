#!/bin/bash
# usage: tools/pmc_passes.sh <out.txt> <kernel substring> -- runs tools/edge_stack_run.py under several --pmc passes
out=$1; filt=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $root/$out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p$i -- python3 $root/tools/edge_stack_run.py > /tmp/pmc$i.log 2>&1
  db=$(find /tmp/pmc$i -name "*.db" | head -1)
  python3 $root/tools/rocpd_pmc.py $db "$filt" >> $root/$out 2>&1
done
