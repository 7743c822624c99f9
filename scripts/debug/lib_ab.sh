#!/bin/bash
# interleaved A/B of the pipelined bench over libraries: scripts/debug/lib_ab.sh OUTDIR REPS STEPS tag=path-or-"shipped" ...
# (libraries built by `python deeppointmap_amd/csrc/build.py --out <lib> [--only a.hip] <flags>`; DPM_LIB selects one)
out=$1; reps=$2; steps=$3; shift 3
mkdir -p $out
for i in $(seq 1 $reps); do
for s in "$@"; do
  tag=${s%%=*}; lib=${s#*=}
  if [ "$lib" = shipped ]; then envs=""; else envs="DPM_LIB=$lib"; fi
  env $envs python bench.py --steps $steps --warmup 10 --allow-knobs --no-extras --cpu-frames 0 > $out/${tag}_$i.json 2> $out/${tag}_$i.err
  python - <<PY
import json
try:
    d = json.loads(open('$out/${tag}_$i.json').read().strip().splitlines()[-1]); g = d.get('parity_gate', {})
    print('$tag $i', d['value'], d.get('ms_per_step'), 'fps us/round', d['roofline'].get('us_per_round'), 'gate', g.get('ok'), g.get('max_dT_m'), g.get('descriptor_max_err'), d.get('error'))
except Exception as e:
    print('$tag $i FAILED', e)
PY
done
done
