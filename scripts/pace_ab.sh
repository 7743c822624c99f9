#!/bin/bash
# Round 4, co-run experiment inside the pipelined bench: libraries whose dense kernels idle n wait states after every
# matrix instruction (-DDPM_MFMA_PACE=n, csrc/dpm_common.h), with and without occupancy caps on the matrix kernels.
#   for p in 3 7 15 31; do python deeppointmap_amd/csrc/build.py --out deeppointmap_amd/libdpm_pace$p.so -DDPM_EXPERIMENT -DDPM_MFMA_PACE=$p; done
#   python deeppointmap_amd/csrc/build.py --out deeppointmap_amd/libdpm_exp.so -DDPM_EXPERIMENT
# usage (GPU box): scripts/pace_ab.sh <outdir>
out=${1:-gpurun_out/pace}; mkdir -p $out
run() {  # tag, env...
  tag=$1; shift
  env "$@" python bench.py --allow-knobs --steps 40 --warmup 5 --cpu-frames 0 --no-extras > $out/$tag.json 2> $out/$tag.err
  python - "$tag" $out/$tag.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"| {sys.argv[1]} | {j['value']:.0f} | {j['ms_per_step']:.3f} | {j['roofline']['us_per_round']:.3f} | {j['roofline_mfma']['avg_launch_ms'] * 1e3:.1f} |")
except Exception as e:
    print(f"| {sys.argv[1]} | failed: {e} |")
PY
}
echo "| library / caps | frames/s | ms per step | sampling us per round | 256->768 projection us (in pipeline) |"
echo "|---|---|---|---|---|"
L=$PWD/deeppointmap_amd
for rep in 1 2; do
run base_$rep DPM_LIB=$L/libdpm_exp.so
for p in 3 7 15 31; do
  run pace${p}_$rep DPM_LIB=$L/libdpm_pace$p.so
done
done
# occupancy caps on the matrix kernels (workgroups per CU) next to the pacing that fits them
run pace7_gemm4 DPM_LIB=$L/libdpm_pace7.so DPM_GEMM_LDS_PAD=23040 DPM_ATT_LDS_PAD=22528
run pace7_gemm2 DPM_LIB=$L/libdpm_pace7.so DPM_GEMM_LDS_PAD=64000 DPM_ATT_LDS_PAD=63488
run pace15_gemm4 DPM_LIB=$L/libdpm_pace15.so DPM_GEMM_LDS_PAD=23040 DPM_ATT_LDS_PAD=22528
run pace3_gemm4 DPM_LIB=$L/libdpm_pace3.so DPM_GEMM_LDS_PAD=23040 DPM_ATT_LDS_PAD=22528
run base_gemm4 DPM_LIB=$L/libdpm_exp.so DPM_GEMM_LDS_PAD=23040 DPM_ATT_LDS_PAD=22528
