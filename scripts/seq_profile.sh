#!/bin/bash
# kernel-by-kernel sequence of one unpipelined benchmark step: scripts/seq_profile.sh <tag>
tag=${1:-seq}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/unpip -o t -- python bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-pipeline --no-extras > $out/unpip.log 2>&1
python scripts/rocprof_summary.py $out/unpip/t_results.db > $out/unpipelined.md
python scripts/kernel_sequence.py $out/unpip/t_results.db > $out/unpipelined_sequence.md
rm -rf $out/unpip
