mkdir -p gpurun_out/r06h
DPM_LIB=$PWD/deeppointmap_amd/csrc/build/libdpm_exp.so timeout 900 python scripts/step_model.py > gpurun_out/r06h/step_model_exp.txt 2>&1; tail -27 gpurun_out/r06h/step_model_exp.txt
