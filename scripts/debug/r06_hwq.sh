mkdir -p gpurun_out/r06l
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --no-extras --cpu-frames 0 > gpurun_out/r06l/$tag.json 2> gpurun_out/r06l/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/r06l/$tag.json').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],d['roofline']['us_per_round'],d['parity_gate']['ok'])"; }
for i in 1 2; do
run default_$i X=1
run hwq8_$i GPU_MAX_HW_QUEUES=8
run hwq2_$i GPU_MAX_HW_QUEUES=2
run hwq16_$i GPU_MAX_HW_QUEUES=16
done
