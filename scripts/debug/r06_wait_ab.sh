mkdir -p gpurun_out/r06ad
run() { tag=$1; shift; python bench.py --steps $S --warmup 10 --no-extras --cpu-frames 0 "$@" > gpurun_out/r06ad/$tag.json 2> gpurun_out/r06ad/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/r06ad/$tag.json').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],d['roofline']['us_per_round'],d['parity_gate']['ok'])"; }
for i in 1 2 3; do
S=60; run wait_60_$i --wait-for-caller-stream 1
S=60; run free_60_$i --wait-for-caller-stream 0
S=20; run wait_20_$i --wait-for-caller-stream 1
S=20; run free_20_$i --wait-for-caller-stream 0
done
