mkdir -p gpurun_out/r06b
rm -f gpurun_out/observed_errors.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06b/pytest.txt; cat gpurun_out/r06b/pytest.txt
cp gpurun_out/observed_errors.log gpurun_out/r06b/ 2>/dev/null
timeout 300 python scripts/margin_table.py > gpurun_out/r06b/margin_default.md 2>&1; tail -25 gpurun_out/r06b/margin_default.md
DPM_GEMM_BF16X3=0 DPM_GEMM_LN_BF16X3=0 timeout 300 python scripts/margin_table.py > gpurun_out/r06b/margin_fp32gemm.md 2>&1; tail -3 gpurun_out/r06b/margin_fp32gemm.md
timeout 600 python scripts/step_model.py > gpurun_out/r06b/step_model.txt 2>&1; tail -22 gpurun_out/r06b/step_model.txt
bash scripts/debug/lib_ab.sh gpurun_out/r06b/ob 2 60 shipped=shipped ob256=$PWD/deeppointmap_amd/csrc/build/libdpm_ob256.so ob512=$PWD/deeppointmap_amd/csrc/build/libdpm_ob512.so
python bench.py --steps 20 --warmup 5 > gpurun_out/r06b/bench_driver_like.json 2> gpurun_out/r06b/bench_driver_like.err; python -c "
import json;d=json.loads(open('gpurun_out/r06b/bench_driver_like.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['parity_gate'])"
