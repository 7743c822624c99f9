#!/bin/bash
# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   kernel trace of the unpipelined and the pipelined bench, and separate PMC passes for
#   FETCH_SIZE / WRITE_SIZE / MFMA counters (never combined with tracing domains other than kernel-trace).
# usage: scripts/collect_profiles.sh <tag>
set -u
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/unpip -o t -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pipeline --no-extras > $out/unpip.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/pip -o t -- python bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-extras > $out/pip.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$c -o p -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-pipeline --no-extras > $out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o p -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-pipeline --no-extras > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_valu -o p -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-pipeline --no-extras > $out/pmc_valu.log 2>&1
python scripts/rocprof_summary.py $out/unpip/t_results.db > $out/unpipelined.md
python scripts/rocprof_summary.py $out/pip/t_results.db > $out/pipelined.md
python scripts/rocprof_pmc.py $out/pmc_FETCH_SIZE/p_results.db $out/pmc_WRITE_SIZE/p_results.db > $out/pmc_fetch_write.md
python scripts/kernel_sequence.py $out/unpip/t_results.db > $out/unpipelined_sequence.md
python scripts/make_pmc_traffic.py $out/pmc_FETCH_SIZE/p_results.db $out/pmc_WRITE_SIZE/p_results.db 64 > $out/pmc_traffic.json
python scripts/evidence_table.py $out/unpip/t_results.db $out/pmc_FETCH_SIZE/p_results.db $out/pmc_WRITE_SIZE/p_results.db $out/pmc_mfma/p_results.db > $out/evidence.md
python scripts/stage_table.py $out/unpip/t_results.db $out/pmc_mfma/p_results.db $out/pmc_valu/p_results.db > $out/stage_table.md
tail -1 $out/unpip.log | cut -c1-200; tail -1 $out/pip.log | cut -c1-200
# gpurun copies at most 64 MiB back: the sqlite traces stay on the box unless asked for
[ "${KEEP_DB:-0}" = 1 ] || rm -rf $out/unpip $out/pip $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_mfma $out/pmc_valu
ls -la $out
