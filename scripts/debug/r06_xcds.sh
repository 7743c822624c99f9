mkdir -p gpurun_out/r06ac
E=$PWD/deeppointmap_amd/csrc/build/libdpm_exp.so
DPM_LIB=$E DPM_FPS_XCDS=2 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k fps 2>&1 | tail -2
for x in 0 4 2 1; do echo "xcds $x:"; DPM_LIB=$E DPM_FPS_XCDS=$x python scripts/fps_algo_bench.py 64 5 | tail -1; done
run() { tag=$1; shift; env DPM_LIB=$E "$@" python bench.py --steps 60 --warmup 10 --no-extras --cpu-frames 0 --allow-knobs > gpurun_out/r06ac/$tag.json 2> gpurun_out/r06ac/$tag.err; python -c "
import json
try:
    d=json.loads(open('gpurun_out/r06ac/$tag.json').read().strip().splitlines()[-1]);print('$tag',d['value'],d['ms_per_step'],d['roofline']['us_per_round'],d['parity_gate']['ok'])
except Exception as e: print('$tag FAILED', e)"; }
for i in 1 2; do
run all_$i DPM_FPS_XCDS=0
run x4_$i DPM_FPS_XCDS=4
run x2_$i DPM_FPS_XCDS=2
run x1_$i DPM_FPS_XCDS=1
done
