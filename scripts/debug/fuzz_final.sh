mkdir -p gpurun_out/r06fuzz
for s in "fuzz_encoder 505 180" "fuzz_ops 506 120" "fuzz_decoder 507 120" "fuzz_pipeline 508 90" "fuzz_misc 509 90" "fuzz_configs 510 90"; do
  set -- $s
  timeout 400 python scripts/$1.py $2 $3 > gpurun_out/r06fuzz/$1.log 2>&1; echo "$1 rc=$?"; tail -4 gpurun_out/r06fuzz/$1.log
done
timeout 300 python scripts/fuzz_path.py 700 900 > gpurun_out/r06fuzz/fuzz_path.log 2>&1; echo "fuzz_path rc=$?"; tail -4 gpurun_out/r06fuzz/fuzz_path.log
