mkdir -p gpurun_out/r05o
run() { tag=$1; shift; python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras "$@" > gpurun_out/r05o/$tag.json 2>gpurun_out/r05o/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05o/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), d['roofline'].get('us_per_round'), d.get('error'))"; }
for i in 1 2; do
run d2_$i
run d3_$i --geometry-depth 3
run d4_$i --geometry-depth 4
done
