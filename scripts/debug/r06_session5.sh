mkdir -p gpurun_out/r06e
R=$PWD/deeppointmap_amd/csrc/build/libdpm_regcl.so
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_system.py -m gpu -q -s 2>&1 | grep -v "^$" | grep "one-pair\|passed\|failed\|Error" | tail -8
DPM_LIB=$R timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k fps 2>&1 | tail -3
for b in 64 8 1; do
  python scripts/fps_algo_bench.py $b 5 2>&1 | tail -1
  DPM_LIB=$R python scripts/fps_algo_bench.py $b 5 2>&1 | tail -1
done
bash scripts/debug/lib_ab.sh gpurun_out/r06e/regcl 2 60 shipped=shipped regcl=$R
timeout 900 python scripts/step_model.py > gpurun_out/r06e/step_model.txt 2>&1; tail -19 gpurun_out/r06e/step_model.txt
