#!/bin/bash
# Co-residency stress (tests/corun_stress.py) over the experimental builds of csrc/build/*.so: which compilation of the
# gather kernels changes results next to bf16 matrix instructions.  Usage (GPU box): scripts/corun_matrix.sh <out file> [iters] [libs...]
out=${1:-gpurun_out/corun_matrix.jsonl}; iters=${2:-300}; shift 2
libs=${@:-$(ls deeppointmap_amd/csrc/build/v*.so)}
mkdir -p $(dirname $out)
python tests/corun_stress.py $iters bf16x3 2>>$out.err | tail -1 | tee -a $out
for l in $libs; do
  DPM_LIB=$PWD/$l timeout 600 python tests/corun_stress.py $iters bf16x3 2>>$out.err | tail -1 | tee -a $out
done
