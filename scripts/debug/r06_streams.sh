mkdir -p gpurun_out/r06aa
run() { tag=$1; q=$2; shift 2; env GPU_MAX_HW_QUEUES=$q python bench.py --steps 60 --warmup 10 --no-extras --cpu-frames 0 "$@" > gpurun_out/r06aa/$tag.json 2> gpurun_out/r06aa/$tag.err; python -c "
import json
try:
    d=json.loads(open('gpurun_out/r06aa/$tag.json').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],d['roofline']['us_per_round'],d['parity_gate']['ok'])
except Exception as e: print('$tag FAILED', e)"; }
for i in 1 2; do
run base_q4_$i 4
run base_q8_$i 8
run split1_q8_$i 8 --feature-split 1
run split2_q8_$i 8 --feature-split 2
run split3_q8_$i 8 --feature-split 3
run fstreams2_q8_$i 8 --feature-streams 2
run split2_q4_$i 4 --feature-split 2
done
