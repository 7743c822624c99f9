mkdir -p gpurun_out/r05n
python scripts/attention_time.py 2>&1 | grep -v amdgpu | head -4
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_attw4.so python scripts/attention_time.py 2>&1 | grep -v amdgpu | head -4
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05n/$tag.json 2>gpurun_out/r05n/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05n/$tag.json').read().strip().splitlines()[-1]); g=d.get('parity_gate',{}); print('$tag', d['value'], d.get('ms_per_step'), g.get('ok'), g.get('max_dT_m'), g.get('descriptor_max_err'), d.get('error'))"; }
for i in 1 2 3; do
run w4_$i DPM_LIB=deeppointmap_amd/csrc/build/libdpm_attw4.so
run base_$i DPM_KV_PLANES=1
done
