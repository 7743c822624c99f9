# round 6: the decoder fuzz over eight seeds with every case above half the pose tolerance dumped (inputs + both results):
# the inputs of tests/golden/make_golden_margin.py
mkdir -p gpurun_out/r06a/cases
for s in 507 601 602 603 604 605 606 607; do
  timeout 300 python scripts/fuzz_decoder.py $s 100 gpurun_out/r06a/cases > gpurun_out/r06a/fuzz_decoder_$s.log 2>&1; echo "seed $s rc=$?"; tail -1 gpurun_out/r06a/fuzz_decoder_$s.log
done
ls gpurun_out/r06a/cases | wc -l
