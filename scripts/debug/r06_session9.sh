mkdir -p gpurun_out/r06i
timeout 900 python scripts/step_model.py > gpurun_out/r06i/step_model.txt 2>&1; tail -30 gpurun_out/r06i/step_model.txt
