#!/bin/bash
# PMC passes of one script (separate passes, kernel-trace only): scripts/pmc_one.sh <tag> <kernel substring> <script> [args]
tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out; rm -f $out/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p -- python "$@" > $out/p$i.log 2>&1
  python scripts/pmc_dump.py "$pat" $out/p$i/p_results.db >> $out/counters.txt 2>&1
  rm -rf $out/p$i
done
cat $out/counters.txt
