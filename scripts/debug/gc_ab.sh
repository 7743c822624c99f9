mkdir -p gpurun_out/r05u
for i in 1 2 3; do
python -c "import gc; gc.disable(); import runpy, sys; sys.argv=['bench.py','--steps','60','--warmup','10','--no-extras']; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nogc', d['value'], d['ms_per_step'])"
python bench.py --steps 60 --warmup 10 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['value'], d['ms_per_step'])"
done
python scripts/host_enqueue_time.py 2>&1 | grep -v amdgpu | tail -3
