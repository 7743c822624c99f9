mkdir -p gpurun_out/r05g
python scripts/infomat_time.py 2>&1 | grep -v amdgpu.ids
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_exp.so DPM_NN1_LANE=0 python scripts/infomat_time.py 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --allow-knobs > gpurun_out/r05g/lane_$i.json 2>gpurun_out/r05g/lane_$i.err
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_exp.so DPM_NN1_LANE=0 python bench.py --steps 60 --warmup 10 --allow-knobs > gpurun_out/r05g/quad_$i.json 2>gpurun_out/r05g/quad_$i.err
done
for f in lane_1 quad_1 lane_2 quad_2; do python -c "import json,sys; d=json.loads(open('gpurun_out/r05g/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d.get('ms_per_step'), d.get('parity_gate',{}).get('ok'), d.get('error'))"; done
