mkdir -p gpurun_out/r06u
B=$PWD/deeppointmap_amd/csrc/build
for v in shipped abl0 abl1 abl2 abl3; do
  if [ $v = shipped ]; then e=""; else e="DPM_LIB=$B/libdpm_$v.so"; fi
  env GPU_MAX_HW_QUEUES=16 STEPS=30 $e timeout 600 python scripts/step_model.py > gpurun_out/r06u/$v.txt 2>&1
  echo "== $v"; grep "^| F | R (samp\|^| first-level sampling only" gpurun_out/r06u/$v.txt
  env $e python scripts/fps_algo_bench.py 64 5 | tail -1
done
