mkdir -p gpurun_out/r06f
timeout 900 python scripts/step_model.py > gpurun_out/r06f/step_model.txt 2>&1; tail -21 gpurun_out/r06f/step_model.txt
timeout 300 python scripts/stage_timeline.py > gpurun_out/r06f/stage_timeline.txt 2>&1; tail -3 gpurun_out/r06f/stage_timeline.txt
bash scripts/collect_profiles.sh r06f_prof > gpurun_out/r06f/collect.log 2>&1; tail -5 gpurun_out/r06f/collect.log
