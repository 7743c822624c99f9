# final evidence of round 6: full GPU suite, driver-like and 60-step lines, two-rank dry run through bench.py's own launcher, profiles
mkdir -p gpurun_out/r06final
rm -f gpurun_out/observed_errors.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r06final/pytest.txt; cat gpurun_out/r06final/pytest.txt
cp gpurun_out/observed_errors.log gpurun_out/r06final/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06final/smoke.txt 2>&1; tail -1 gpurun_out/r06final/smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06final/bench_driver_like.json 2> gpurun_out/r06final/bench_driver_like.err
python bench.py --steps 60 --warmup 10 --cpu-frames 0 > gpurun_out/r06final/bench_60_steps.json 2> gpurun_out/r06final/bench_60.err
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --cpu-frames 0 > gpurun_out/r06final/bench_2rank_gloo_dryrun.json 2> gpurun_out/r06final/bench_2rank.err
for f in bench_driver_like bench_60_steps bench_2rank_gloo_dryrun; do python -c "
import json;d=json.loads(open('gpurun_out/r06final/$f.json').read().strip().splitlines()[-1]);print('$f',d['value'],d['ms_per_step'],d.get('sustained',{}).get('value'),d['parity_gate'].get('ok'),d.get('latency_mode_ms_per_frame'),d.get('rank0_serial_ms'),d.get('ranks',{}).get('world_size'))"; done
bash scripts/collect_profiles.sh r06final_prof > gpurun_out/r06final/collect.log 2>&1; tail -3 gpurun_out/r06final/collect.log
python scripts/reg_shapes_bench.py > gpurun_out/r06final/reg_shapes.txt 2>&1; tail -4 gpurun_out/r06final/reg_shapes.txt
