mkdir -p gpurun_out/r06s
B=$PWD/deeppointmap_amd/csrc/build
run() { tag=$1; lib=$2; shift 2; env GPU_MAX_HW_QUEUES=8 $lib python bench.py --steps 60 --warmup 10 --no-extras --cpu-frames 0 --allow-knobs "$@" > gpurun_out/r06s/$tag.json 2> gpurun_out/r06s/$tag.err; python -c "
import json;d=json.loads(open('gpurun_out/r06s/$tag.json').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],d['roofline']['us_per_round'],d['parity_gate']['ok'])"; }
for i in 1 2; do
run q8_shipped_d2_$i X=1
run q8_shipped_d3_$i X=1 --geometry-depth 3
run q8_shipped_d4_$i X=1 --geometry-depth 4
run q8_prio1_d3_$i DPM_LIB=$B/libdpm_prio1.so --geometry-depth 3
run q8_prio1_d4_$i DPM_LIB=$B/libdpm_prio1.so --geometry-depth 4
done
