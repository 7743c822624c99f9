#!/bin/bash
# alternating A/B of the pipelined bench under environment settings: scripts/ab_bench.sh "VAR=1" "VAR=0" ...
# The kernels' switches exist only in an experimental build: python deeppointmap_amd/csrc/build.py --out /tmp/libdpm_exp.so -DDPM_EXPERIMENT
# and add DPM_LIB=/tmp/libdpm_exp.so to every setting; bench.py needs --allow-knobs to run with any DPM_* variable set.
for rep in 1 2; do
for setting in "$@"; do
  env $setting python bench.py --allow-knobs --steps 60 --warmup 5 --cpu-frames 0 --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$setting', j['value'], j['ms_per_step'], j['roofline']['us_per_round'])"
done
done
