#!/bin/bash
# PMC study of the attention kernel (128 sequences of 256 tokens; separate passes, kernel-trace only) -> gpurun_out/attention_pmc/counters.txt
set -u
R=${1:-32768}; K=${2:-256}; N=${3:-768}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/attention_pmc; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_LEVEL_WAVES SQ_WAVES SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $out/p$i -o p -- python scripts/attention_one.py 128 12 > $out/p$i.log 2>&1
  python scripts/pmc_dump.py attention_kernel $out/p$i/p_results.db >> $out/counters.txt 2>&1
  rm -rf $out/p$i
done
cat $out/counters.txt
