mkdir -p gpurun_out/r06c
rm -f gpurun_out/observed_errors.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r06c/pytest.txt; cat gpurun_out/r06c/pytest.txt
cp gpurun_out/observed_errors.log gpurun_out/r06c/ 2>/dev/null
timeout 600 python scripts/step_model.py > gpurun_out/r06c/step_model.txt 2>&1; tail -20 gpurun_out/r06c/step_model.txt
timeout 300 python scripts/stage_timeline.py > gpurun_out/r06c/stage_timeline.txt 2>&1; tail -12 gpurun_out/r06c/stage_timeline.txt
timeout 200 python scripts/infomat_time.py > gpurun_out/r06c/infomat_new.txt 2>&1; tail -5 gpurun_out/r06c/infomat_new.txt
DPM_LIB=$PWD/deeppointmap_amd/csrc/build/libdpm_base.so timeout 200 python scripts/infomat_time.py > gpurun_out/r06c/infomat_base.txt 2>&1; tail -5 gpurun_out/r06c/infomat_base.txt
bash scripts/debug/lib_ab.sh gpurun_out/r06c/nn1 2 60 base=$PWD/deeppointmap_amd/csrc/build/libdpm_base.so new=shipped
