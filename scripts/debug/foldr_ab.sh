mkdir -p gpurun_out/r05t
python scripts/debug/enc_err_modes.py 50 2>&1 | grep -v amdgpu
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05t/$tag.json 2>gpurun_out/r05t/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05t/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), g.get('max_dT_m'), g.get('descriptor_max_err'), d.get('error'))"; }
for i in 1 2; do
run all_$i DPM_FOLD_MIN_RADIUS=0
run r01_$i DPM_FOLD_MIN_RADIUS=0.15
run r02_$i DPM_FOLD_MIN_RADIUS=0.3
run none_$i DPM_FOLD_MIN_RADIUS=100
done
