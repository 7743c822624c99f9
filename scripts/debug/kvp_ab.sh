mkdir -p gpurun_out/r05l
python -m pytest tests/test_gpu_decoder.py -q -m gpu -x 2>&1 | tail -8
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05l/$tag.json 2>gpurun_out/r05l/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05l/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), g.get('max_dT_m'), g.get('descriptor_max_err'), d.get('error'))"; }
for i in 1 2 3; do
run planes_$i DPM_KV_PLANES=1
run fp32_$i DPM_KV_PLANES=0
done
python scripts/reg_shapes_bench.py 2>&1 | tail -3
DPM_KV_PLANES=0 python scripts/reg_shapes_bench.py 2>&1 | tail -3
