mkdir -p gpurun_out/r05q
python scripts/infomat_time.py 2>&1 | grep -v amdgpu.ids
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_head.so python scripts/infomat_time.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -2
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05q/$tag.json 2>gpurun_out/r05q/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05q/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), g.get('max_dT_m'), d.get('error'))"; }
for i in 1 2 3; do
run new_$i DPM_KV_PLANES=1
run head_$i DPM_LIB=deeppointmap_amd/csrc/build/libdpm_head.so
done
