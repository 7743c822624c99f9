mkdir -p gpurun_out/r05s
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_corun_stress.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -3
python scripts/debug/enc_err_modes.py 60 2>&1 | grep -v amdgpu
python scripts/gather_time.py 2>&1 | grep -v amdgpu | tail -1
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_head.so python scripts/gather_time.py 2>&1 | grep -v amdgpu | tail -1
FULL_CONFIG=1 timeout 400 python scripts/fuzz_encoder.py 701 240 2>&1 | tail -1
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05s/$tag.json 2>gpurun_out/r05s/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05s/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), g.get('max_dT_m'), g.get('descriptor_max_err'), d.get('error'))"; }
for i in 1 2; do
run new_$i DPM_KV_PLANES=1
run head_$i DPM_LIB=deeppointmap_amd/csrc/build/libdpm_head.so
done
