#!/bin/bash
# per-kernel totals of one unpipelined step under environment settings: scripts/ab_seq.sh <grep pattern> "VAR=1" "VAR=2" ...
pat=$1; shift
for setting in "$@"; do
  env $setting bash scripts/seq_profile.sh ab_$$ > /dev/null 2>&1
  echo "== $setting"; grep "$pat" gpurun_out/ab_$$/unpipelined.md | cut -c1-110
done
rm -rf gpurun_out/ab_$$
