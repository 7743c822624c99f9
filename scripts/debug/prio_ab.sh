mkdir -p gpurun_out/r05h
python -c "import torch; print(torch.cuda.Stream.priority_range())"
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --allow-knobs --no-extras > gpurun_out/r05h/$tag.json 2>gpurun_out/r05h/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/r05h/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d.get('ms_per_step'), d.get('parity_gate',{}).get('ok'), d.get('error'))"; }
for i in 1 2; do
run base_$i DPM_PRIO_GEO=0
run reghi_$i DPM_PRIO_REG=-1
run geohi_$i DPM_PRIO_GEO=-1
run bothhi_$i DPM_PRIO_GEO=-1 DPM_PRIO_REG=-1
done
