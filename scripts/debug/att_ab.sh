mkdir -p gpurun_out/r05f
for i in 1 2 3; do
python bench.py --steps 60 --warmup 10 --allow-knobs > gpurun_out/r05f/b8_$i.json 2>gpurun_out/r05f/b8_$i.err
DPM_LIB=deeppointmap_amd/csrc/build/libdpm_att4.so python bench.py --steps 60 --warmup 10 --allow-knobs > gpurun_out/r05f/b4_$i.json 2>gpurun_out/r05f/b4_$i.err
done
for f in b8_1 b4_1 b8_2 b4_2 b8_3 b4_3; do python -c "import json,sys; d=json.loads(open('gpurun_out/r05f/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d.get('ms_per_step'), d['parity_gate'].get('ok'))"; done
