mkdir -p gpurun_out/r05i
python -m pytest tests -q -m gpu -x 2>&1 | tail -5
python scripts/gather_time.py 2>&1 | grep -v amdgpu.ids | tail -12
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05i/$tag.json 2>gpurun_out/r05i/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05i/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d.get('ms_per_step'), d.get('parity_gate'), d.get('error'))"; }
for i in 1 2; do
run centred_$i DPM_FOLD_GATHER=1
run plain_$i DPM_CENTRED_GATHER=0
done
