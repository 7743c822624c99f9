mkdir -p gpurun_out/r06d
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_system.py tests/test_gpu_multiagent.py tests/test_gpu_multirank.py tests/test_gpu_cloud.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06d/pytest.txt; cat gpurun_out/r06d/pytest.txt
timeout 900 python scripts/step_model.py > gpurun_out/r06d/step_model.txt 2>&1; tail -20 gpurun_out/r06d/step_model.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06d/bench_driver_like.json 2> gpurun_out/r06d/bench_driver_like.err; python -c "
import json;d=json.loads(open('gpurun_out/r06d/bench_driver_like.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d.get('sustained'),d['parity_gate']['ok'])"
