#!/bin/bash
# kernel trace of one script: scripts/prof_one.sh <tag> <python script> [args]; prints the per-kernel table
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/tr -o t -- python "$@" > $out/run.log 2>&1
python scripts/rocprof_summary.py $out/tr/t_results.db > $out/kernels.md
cat $out/kernels.md | cut -c1-160
rm -rf $out/tr
